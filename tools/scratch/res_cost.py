"""What does the residual read cost a packed conv3 launch?  (same launch with and without the residual pointer, graph-replayed)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from stabnet_amd import ops, _lib
from stabnet_amd._tensor import ptr, stream_ptr
dev = torch.device("cuda:0")
L = _lib.lib()
def bench(N, H, W, Cin, Cout, with_res, packed=True):
    x = torch.randn(N, H, W, Cin, device=dev) * 0.5
    w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05
    r = torch.randn(N, H, W, Cout, device=dev) if with_res else None
    y = torch.empty(N, H, W, Cout, device=dev)
    img = torch.empty(int(L.stabnet_conv_weight_image_floats(Cout, 1, 1, Cin)), device=dev)
    _lib.call("stabnet_conv_weight_split_image", ptr(w), Cout, 1, 1, Cin, ptr(img), stream_ptr(dev), device=dev)
    wsb = max(int(L.stabnet_conv2d_workspace_bytes(N, H, W, Cin, Cout, 1, 1, 1, 0)), 4)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    def call():
        if packed:
            _lib.call("stabnet_conv2d_fwd_packed", ptr(x), ptr(w), ptr(img), 0, 0, 0, ptr(r), H if with_res else 0, W if with_res else 0, 1, 0, 0, ptr(y),
                      N, H, W, Cin, Cout, 1, 1, 1, 0, 0, 1, ptr(ws), wsb, stream_ptr(dev), device=dev)
        else:
            _lib.call("stabnet_conv2d_fwd_ex", ptr(x), ptr(w), 0, 0, 0, ptr(r), H if with_res else 0, W if with_res else 0, 1, 0, 0, ptr(y),
                      N, H, W, Cin, Cout, 1, 1, 1, 0, 0, ptr(ws), wsb, stream_ptr(dev), device=dev)
    call(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): call()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / 400
for (N, H, W, Cin, Cout) in [(1, 45, 80, 256, 1024), (1, 90, 160, 128, 512), (1, 180, 320, 64, 256), (1, 23, 40, 512, 2048)]:
    a, b = bench(N, H, W, Cin, Cout, False), bench(N, H, W, Cin, Cout, True)
    c, d = bench(N, H, W, Cin, Cout, False, False), bench(N, H, W, Cin, Cout, True, False)
    print("M=%d K=%d N=%d: packed %.1f us without / %.1f us with residual; f32 MFMA %.1f / %.1f" % (N * H * W, Cin, Cout, a, b, c, d))
