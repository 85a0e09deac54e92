import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from stabnet_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
for (N, H, W, Cin, Cout) in ((16, 72, 128, 64, 64), (16, 36, 64, 128, 128), (16, 18, 32, 256, 256)):
    Ho, Wo = H // 2, W // 2
    dy = torch.randn(N, Ho, Wo, Cout, device=dev); w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05
    dx = torch.empty(N, H, W, Cin, device=dev)
    nb = L.stabnet_conv2d_dgrad_workspace_bytes(N, H, W, Cin, Cout, 3, 3, 2, 1)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    def run():
        _lib.call("stabnet_conv2d_dgrad", dy.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, N, H, W, Cin, Cout, 3, 3, 2, 1, ws.data_ptr(), nb, st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print("dx %dx%dx%dx%d <- Cout %d: %.1f us per call (pack + dgrad [+ reduce])" % (N, H, W, Cin, Cout, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
