#!/usr/bin/env python3
"""Measure every (tile, split-K) choice for every convolution shape of the regressor and print the best per shape
(the table in csrc/conv.hip is generated from this output).
  python tools/autotune.py --height 720 --width 1280 --batch 1 [--dgrad]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stabnet_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=720); ap.add_argument("--width", type=int, default=1280)
ap.add_argument("--batch", type=int, default=1); ap.add_argument("--reps", type=int, default=12)
ap.add_argument("--dgrad", action="store_true", help="also tune the dgrad convolutions (training)")
a = ap.parse_args()
L = _lib.lib()
dev = torch.device("cuda:0")


def net_convs(N, H, W):
    """(N,H,W,Cin,Cout,k,stride,pad) of every forward conv; dgrad convs as equivalent forward shapes when asked."""
    out = [(N, H, W, 16, 64, 7, 2, 3)]
    h, w = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    h, w = -(-h // 2), -(-w // 2)
    cin = 64
    for depth, dbn, units, bstride in ((256, 64, 3, 2), (512, 128, 4, 2), (1024, 256, 6, 2), (2048, 512, 3, 1)):
        for u in range(1, units + 1):
            s = bstride if u == units else 1
            ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
            if cin != depth:
                out.append((N, h, w, cin, depth, 1, 1, 0))
            out.append((N, h, w, cin, dbn, 1, 1, 0))
            out.append((N, h, w, dbn, dbn, 3, s, 1))
            out.append((N, ho, wo, dbn, depth, 1, 1, 0))
            if a.dgrad:      # dgrad of conv(Cin->Cout, k, s) at input HxW == forward conv (Cout->Cin, k, 1) over HxW (s=1 only)
                out.append((N, ho, wo, depth, dbn, 1, 1, 0))
                if s == 1:
                    out.append((N, h, w, dbn, dbn, 3, 1, 1))
                out.append((N, h, w, dbn, cin, 1, 1, 0))
                if cin != depth:
                    out.append((N, h, w, depth, cin, 1, 1, 0))
            h, w, cin = ho, wo, depth
    seen, uniq = set(), []
    for c in out:
        if c not in seen:
            seen.add(c); uniq.append(c)
    return uniq


def time_conv(shape, tile, sk):
    N, H, W, Cin, Cout, k, s, p = shape
    x = torch.randn(N, H, W, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) * (2.0 / (k * k * Cin)) ** 0.5
    sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
    L.stabnet_conv_tuning_override(tile, sk)
    try:
        for _ in range(2):
            ops.conv2d(x, w, None, sc, sh, None, 1, s, p, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            ops.conv2d(x, w, None, sc, sh, None, 1, s, p, False)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps * 1e3
    finally:
        L.stabnet_conv_tuning_override(-1, -1)


print("# H=%d W=%d N=%d" % (a.height, a.width, a.batch))
tot_best = tot_def = 0.0
for shape in net_convs(a.batch, a.height, a.width):
    N, H, W, Cin, Cout, k, s, p = shape
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    M, K = N * Ho * Wo, k * k * Cin
    steps = k * k * (Cin // (32 if Cin % 32 == 0 else 16))
    t_def = time_conv(shape, -1, -1)
    res = []
    for tile in (0, 1, 2):
        if (tile == 0 and Cout < 128):
            continue
        for sk in (1, 2, 3, 4, 6, 8, 12, 16):
            if sk > 1 and (steps // sk < 4):
                continue
            res.append((time_conv(shape, tile, sk), tile, sk))
    best = min(res)
    tot_best += best[0]; tot_def += t_def
    fl = 2.0 * M * K * Cout
    print("    {%7d, %5d, %5d, %d, %d, %2d},   // default %6.1f us -> best %6.1f us (%5.1f TF)  top3: %s" % (
        M, Cout, K, k, best[1], best[2], t_def, best[0], fl / best[0] / 1e6,
        " ".join("t%d/s%d=%.1f" % (t, sk, us) for us, t, sk in sorted(res)[:3])))
print("# total default %.1f us, best %.1f us" % (tot_def, tot_best))
