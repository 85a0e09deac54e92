"""Hypothesis test: do two half-size dependent chains on two streams hide each other's per-launch fill/drain?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from stabnet_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()

def conv(x, w, y, res, M, Cin, Cout, ws, stream):
    # 1x1 conv over a [1, 1, M, Cin] "image"
    _lib.call("stabnet_conv2d_fwd_ex", x.data_ptr(), w.data_ptr(), 0, 0, 0, res.data_ptr() if res is not None else 0, 1, M, 1, 0, 0,
              y.data_ptr(), 1, 1, M, Cin, Cout, 1, 1, 1, 0, 0, ws.data_ptr(), ws.numel(), stream.cuda_stream)

def run(M, C1, C2, pairs, mode):
    x = torch.randn(M, C1, device=dev); y = torch.empty(M, C2, device=dev); z = torch.empty(M, C1, device=dev)
    r = torch.randn(M, C2, device=dev)
    w1 = torch.randn(C2, C1, device=dev) * 0.05; w2 = torch.randn(C1, C2, device=dev) * 0.05
    wsb = 64 << 20
    ws = [torch.empty(wsb, dtype=torch.uint8, device=dev) for _ in range(2)]
    s2 = torch.cuda.Stream(device=dev)
    h = (M // 2 // 64) * 64
    def body(main):
        if mode == "single":
            a, b = x, z
            for _ in range(pairs):
                conv(a, w1, y, r, M, C1, C2, ws[0], main)
                conv(y, w2, b, None, M, C2, C1, ws[0], main)
                a = b
        else:
            ev = torch.cuda.Event(); ev.record(main); s2.wait_event(ev)
            for _ in range(pairs):
                for (st, lo, n, k) in ((main, 0, h, 0), (s2, h, M - h, 1)):
                    conv(x[lo:lo + n] if _ == 0 else z[lo:lo + n], w1, y[lo:lo + n], r[lo:lo + n], n, C1, C2, ws[k], st)
                    conv(y[lo:lo + n], w2, z[lo:lo + n], None, n, C2, C1, ws[k], st)
                if mode == "join":
                    e2 = torch.cuda.Event(); e2.record(s2); main.wait_event(e2)
                    e3 = torch.cuda.Event(); e3.record(main); s2.wait_event(e3)
            e2 = torch.cuda.Event(); e2.record(s2); main.wait_event(e2)
    main = torch.cuda.current_stream(dev)
    body(main); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(torch.cuda.current_stream(dev))
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3 / pairs

for (M, C1, C2) in ((57600, 64, 256), (14400, 128, 512), (3600, 256, 1024), (920, 512, 2048)):
    res = {m: run(M, C1, C2, 8, m) for m in ("single", "two", "join")}
    fl = 2.0 * M * C1 * C2 * 2
    print("M=%d %d<->%d : us per (conv3+conv1) pair: single %.1f | two independent chains %.1f | two chains joined per pair %.1f   (%.0f TF single)"
          % (M, C1, C2, res["single"], res["two"], res["join"], fl / res["single"] / 1e6), flush=True)
