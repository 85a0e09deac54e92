"""Feasibility probe: does an independent, stem-sized convolution on a SECOND stream overlap the frame's layer chain?
(Idea: 11 of the stem's 13 input channels of frame t+1 do not depend on frame t's output -- their part of the stem could run
beside frame t.)  Measures frames/s alone, side convolutions/s alone, and both at once."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from stabnet_amd import ops, synthetic
from stabnet_amd.config import Config
from stabnet_amd.deploy import StabNetStream

dev = torch.device("cuda:0")
H, W = 720, 1280
cfg = Config(height=H, width=W)
P = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
clip = torch.from_numpy(synthetic.make_clip(H, W, 8, seed=1)).to(dev)
fr = [clip[t:t + 1].contiguous() for t in range(8)]
s = StabNetStream(P, H, W, cfg, streams=1, device=dev, use_graph=True)
s.start(fr[0])
for i in range(10):
    s.step(fr[i % 8])
torch.cuda.synchronize()
# the side work: a 3x3 conv, M = 360*640 = 230400, Cin = Cout = 64 (17 GF; the stem is 19.8 GF), ring kernel, on its own stream
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((1, 360, 640, 64)).astype(np.float32)).to(dev)
w = torch.from_numpy(ops.pack_conv_weight((rng.standard_normal((3, 3, 64, 64)) * 0.05).astype(np.float32))).to(dev)
side = torch.cuda.Stream(device=dev)
wgs = os.environ.get("STABNET_CONV_RING_WGS_PER_CU", "3")
def run_side(n):
    with torch.cuda.stream(side):
        for _ in range(n):
            ops.conv2d(x, w, None, None, None, None, 1, 1, 1, False)
run_side(3); torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for i in range(N): s.step(fr[i % 8])
torch.cuda.synchronize(); t_frames = time.perf_counter() - t0
t0 = time.perf_counter(); run_side(N); torch.cuda.synchronize(); t_side = time.perf_counter() - t0
t0 = time.perf_counter()
for i in range(N):
    run_side(1)
    s.step(fr[i % 8])
torch.cuda.synchronize(); t_both = time.perf_counter() - t0
print("ring WGs/CU=%s  frame alone %.3f ms | side conv alone %.3f ms | both per iteration %.3f ms (sum %.3f, max %.3f)" % (
    wgs, 1e3 * t_frames / N, 1e3 * t_side / N, 1e3 * t_both / N, 1e3 * (t_frames + t_side) / N, 1e3 * max(t_frames, t_side) / N))
