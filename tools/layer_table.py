#!/usr/bin/env python3
"""Per-launch table of one deploy frame (kernel, GEMM shape, us, TFLOP/s, GB/s) from the in-library HIP-event profiler.
  python tools_layer_table.py [--height 720 --width 1280 --reps 20]"""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabnet_amd import synthetic
from stabnet_amd.config import Config
from stabnet_amd.deploy import Profiler, StabNetStream

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=720); ap.add_argument("--width", type=int, default=1280)
ap.add_argument("--streams", type=int, default=1); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--bf16", action="store_true", help="the secondary bf16-operand mode (shows each layer's non-MFMA floor)")
ap.add_argument("--mode", type=int, default=0, help="conv operand mode (stabnet_net_set_bf16_operands): 2 / 3 split, 4 packed split")
a = ap.parse_args()
cfg = Config(height=a.height, width=a.width)
P = synthetic.make_params(cfg, 0, 0.2)
clip = torch.from_numpy(synthetic.make_clip(a.height, a.width, 4, 1234)).cuda()
s = StabNetStream(P, a.height, a.width, cfg, streams=a.streams, operand_mode=(1 if a.bf16 else a.mode))
fr = [clip[t:t+1].expand(a.streams, a.height, a.width).contiguous() for t in range(4)]
s.start(fr[0])
for i in range(5): s.step(fr[i % 4])
prof = Profiler(a.reps * 200)
for i in range(a.reps): s.step(fr[i % 4], prof)
recs = prof.records_with_shapes()
per = len(recs) // a.reps
tot = 0.0
print("%3s %-42s %8s %6s %6s %3s %9s %8s %8s" % ("#", "kernel", "M", "N", "K", "sk", "us", "TFLOP/s", "GB/s"))
for j in range(per):
    rs = [recs[r * per + j] for r in range(a.reps)]
    ms = float(np.median([r[1] for r in rs])); name, _, fl, by, shp = rs[0]
    tot += ms
    print("%3d %-42s %8d %6d %6d %3d %9.1f %8.1f %8.0f" % (j, name, shp[0], shp[1], shp[2], shp[3], ms * 1e3,
          fl / (ms * 1e-3) / 1e12 if ms > 0 else 0, by / (ms * 1e-3) / 1e9 if ms > 0 else 0))
print("sum of kernel times: %.3f ms" % tot)
