"""Split-operand mode (exact f32 products on the bf16 matrix pipe): theta vs the oracle and vs the f32-MFMA path, 720p frame time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import stabnet_oracle as O
from stabnet_amd import synthetic
from stabnet_amd.config import Config
from stabnet_amd.regressor import Regressor
from stabnet_amd.deploy import StabNetStream
dev = torch.device("cuda:0")
import re
from stabnet_amd import _lib
tab = os.environ.get("PROBE_TABLE")
if tab:
    L = _lib.lib()
    for m in re.finditer(r"\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\},", open(tab).read()):
        v = [int(x) for x in m.groups()]
        if v[0] > 0:
            L.stabnet_conv_tuning_table_set(*v)
    print("runtime split-K table:", tab)
modes = [0, 4]
for (N, H, W) in [(2, 64, 96), (1, 288, 512)]:
    cfg, ocfg = Config(height=H, width=W), O.Config(height=H, width=W)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
    x, _ = synthetic.make_stack(cfg, N, H, W, seed=3)
    xt = torch.from_numpy(x).to(dev)
    ref, _, _ = O.get_resnet(x, P, ocfg)
    ref64 = None
    th = {}
    for m in modes:
        th[m] = Regressor(P, N, H, W, cfg, operand_mode=m)(xt).cpu().numpy()
    print("%dx%dx%d theta scale %.3f" % (N, H, W, np.abs(ref).max()))
    for m in modes:
        print("   mode %d: max |theta - oracle| %.3e   max |theta - mode0| %.3e" % (m, np.abs(th[m] - ref).max(), np.abs(th[m] - th[0]).max()))
if len(sys.argv) > 1 and sys.argv[1] == "parity":
    sys.exit(0)
H, W = 720, 1280
cfg = Config(height=H, width=W)
P = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
clip = torch.from_numpy(synthetic.make_clip(H, W, 16, seed=1234)).to(dev)
for m in modes:
    st = StabNetStream(P, H, W, cfg, streams=1, device=dev, use_graph=True, operand_mode=m)
    st.start(clip[0:1])
    t = 1
    for _ in range(30):
        st.step(clip[t % 16:t % 16 + 1]); t += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        st.step(clip[t % 16:t % 16 + 1]); t += 1
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print("720p mode %d: %.4f ms/frame = %.1f frames/s   theta[0,:3] %s" % (m, ms, 1e3 / ms, st.theta[0, :3].cpu().numpy()))
    del st
