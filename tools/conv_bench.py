#!/usr/bin/env python3
"""Single-convolution microbenchmark through the C ABI (for rocprofv3 / tile tuning).
  python tools_conv_bench.py N H W Cin Cout k stride pad [reps] [prologue]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stabnet_amd import ops
a = [int(v) for v in sys.argv[1:9]]
N, H, W, Cin, Cout, k, stride, pad = a
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 50
pro = int(sys.argv[10]) if len(sys.argv) > 10 else 1
dev = torch.device("cuda:0")
x = torch.randn(N, H, W, Cin, device=dev)
w = torch.randn(Cout, k, k, Cin, device=dev) * (2.0 / (k * k * Cin)) ** 0.5
sc = torch.rand(Cin, device=dev) + 0.5 if pro else None
sh = torch.randn(Cin, device=dev) * 0.1 if pro else None
for _ in range(5):
    y = ops.conv2d(x, w, None, sc, sh, None, 1, stride, pad, False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    y = ops.conv2d(x, w, None, sc, sh, None, 1, stride, pad, False)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
M = y.shape[0] * y.shape[1] * y.shape[2]
fl = 2.0 * M * k * k * Cin * Cout
print("M=%d N=%d K=%d  %.1f us  %.1f TFLOP/s" % (M, Cout, k * k * Cin, ms * 1e3, fl / ms / 1e9))
