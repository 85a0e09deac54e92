import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, numpy as np
from stabnet_amd import ops, _lib
dev = torch.device("cuda:0")
def t(N, H, W, Cin, Cout, k=1, reps=200, splitk=None, residual=False):
    x = torch.randn(N, H, W, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    r = torch.randn(N, H, W, Cout, device=dev) if residual else None
    if splitk is not None:
        _lib.lib().stabnet_conv_tuning_override(-1, splitk)
    for _ in range(5):
        y = ops.conv2d(x, w, None, None, None, r, 1, 1, k // 2, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = ops.conv2d(x, w, None, None, None, r, 1, 1, k // 2, False)
    e1.record(); torch.cuda.synchronize()
    _lib.lib().stabnet_conv_tuning_override(-1, -1)
    return e0.elapsed_time(e1) / reps * 1e3
# NOTE ops.conv2d allocates y + workspace per call (torch caching allocator: cheap but host-side ~10 us) -> host-bound risk.
for (H, W, Cout, label) in ((60, 60, 1024, "M=3600 N=1024"), (60, 60, 256, "M=3600 N=256"), (240, 240, 64, "M=57600 N=64"), (240, 240, 256, "M=57600 N=256"), (120, 120, 512, "M=14400 N=512")):
    row = []
    for Cin in (32, 64, 128, 256, 512, 1024, 2048):
        us = t(1, H, W, Cin, Cout, splitk=1)
        fl = 2.0 * H * W * Cin * Cout
        row.append("K=%d: %.1fus %.0fTF" % (Cin, us, fl / us / 1e6))
    print(label, " | ".join(row), flush=True)
