import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from stabnet_amd import synthetic
from stabnet_amd.config import Config
from stabnet_amd.deploy import ClipPipeline, StabNetStream
H, W, T = 720, 1280, 120
dev = torch.device("cuda", 0)
cfg = Config(height=H, width=W)
params = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
base = synthetic.make_clip(H, W, 40, seed=1234).astype(np.float32)
grey = np.ascontiguousarray(base[np.arange(T) % len(base)])
bgr = np.ascontiguousarray(np.repeat(((grey + 0.5) * 255).clip(0, 255).astype(np.uint8)[..., None], 3, axis=3))
consume = int(os.environ.get("CONSUME", "1"))
st = StabNetStream(params, H, W, cfg, device=dev, use_graph=True)
pipe = ClipPipeline(st, colour=True, slots=3)
pipe.run(grey[:8], bgr[:8], sink=lambda r: None)
got_c, got_o = np.zeros((T, H, W, 3), np.uint8), np.zeros((T, H, W), np.uint8)
def sink(r):
    if consume:
        np.copyto(got_c[r["t"]], r["bgr"]); np.copyto(got_o[r["t"]], r["output"])
t0 = time.perf_counter(); pipe.run(grey, bgr, sink=sink); dt = time.perf_counter() - t0
print("fps", (T - 1) / dt, file=sys.stderr)
