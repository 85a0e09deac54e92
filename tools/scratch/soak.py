"""Soak: 20 000 frames at 720p in conv operand mode 4 (hipGraph replay), frames/s per 2 500-frame window, outputs finite throughout."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stabnet_amd import synthetic
from stabnet_amd.config import Config
from stabnet_amd.deploy import StabNetStream
H, W = 720, 1280
dev = torch.device("cuda:0")
cfg = Config(height=H, width=W)
P = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
clip = torch.from_numpy(synthetic.make_clip(H, W, 32, seed=1234)).to(dev)
for mode in (4, 0):
    st = StabNetStream(P, H, W, cfg, streams=1, device=dev, use_graph=True, operand_mode=mode)
    st.start(clip[0:1])
    t = 1
    for _ in range(50):
        st.step(clip[t % 32:t % 32 + 1]); t += 1
    rates = []
    for w in range(8 if mode == 4 else 2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2500):
            st.step(clip[t % 32:t % 32 + 1]); t += 1
        torch.cuda.synchronize()
        rates.append(2500 / (time.perf_counter() - t0))
        assert torch.isfinite(st.theta).all() and torch.isfinite(st.out_img).all() and torch.isfinite(st.frames_ring).all()
    print("operand mode %d: frames/s per 2500-frame window: %s" % (mode, " ".join("%.1f" % r for r in rates)), flush=True)
    del st
