"""Hypothesis test (training): wgrad(L) and dgrad(L) only share dy(L).  Does running the wgrad launches on a second stream, beside
the dgrad chain, fill the per-launch fill / drain / tail bubbles?  Best case for the idea: no joins at all inside the timed region."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from stabnet_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
# (N, H, W, Cin, Cout, K, stride, pad): the pair batch (2 x 8 samples) of training at 288x512
LAYERS = [(16, 72, 128, 64, 256, 1, 1, 0), (16, 72, 128, 64, 64, 3, 1, 1), (16, 72, 128, 256, 64, 1, 1, 0),
          (16, 36, 64, 128, 128, 3, 1, 1), (16, 36, 64, 128, 512, 1, 1, 0), (16, 36, 64, 512, 128, 1, 1, 0),
          (16, 18, 32, 256, 256, 3, 1, 1), (16, 18, 32, 256, 1024, 1, 1, 0), (16, 18, 32, 1024, 256, 1, 1, 0),
          (16, 9, 16, 512, 512, 3, 1, 1), (16, 9, 16, 512, 2048, 1, 1, 0), (16, 9, 16, 2048, 512, 1, 1, 0)]
bufs = []
for (N, H, W, Cin, Cout, K, s, p) in LAYERS:
    x = torch.randn(N, H, W, Cin, device=dev); dy = torch.randn(N, H, W, Cout, device=dev)
    w = torch.randn(Cout, K, K, Cin, device=dev) * 0.05; dw = torch.zeros_like(w); dx = torch.empty_like(x)
    sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
    nb_w = L.stabnet_conv2d_wgrad_workspace_bytes(N, H, W, Cin, Cout, K, K, s, p)
    nb_d = L.stabnet_conv2d_dgrad_workspace_bytes(N, H, W, Cin, Cout, K, K, s, p)
    bufs.append((x, dy, w, dw, dx, sc, sh, torch.empty(nb_w, dtype=torch.uint8, device=dev), nb_w,
                 torch.empty(max(nb_d, 16), dtype=torch.uint8, device=dev), nb_d))

def dgrad(i, st):
    N, H, W, Cin, Cout, K, s, p = LAYERS[i]
    x, dy, w, dw, dx, sc, sh, wsw, nbw, wsd, nbd = bufs[i]
    _lib.call("stabnet_conv2d_dgrad", dy.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, N, H, W, Cin, Cout, K, K, s, p, wsd.data_ptr(), nbd, st.cuda_stream)
def wgrad(i, st):
    N, H, W, Cin, Cout, K, s, p = LAYERS[i]
    x, dy, w, dw, dx, sc, sh, wsw, nbw, wsd, nbd = bufs[i]
    _lib.call("stabnet_conv2d_wgrad", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), sc.data_ptr(), sh.data_ptr(), N, H, W, Cin, Cout, K, K, s, p,
              wsw.data_ptr(), nbw, st.cuda_stream)

main = torch.cuda.current_stream(dev)
s2 = torch.cuda.Stream(device=dev)
def run(mode, reps=10):
    def body():
        if mode == "two":
            ev = torch.cuda.Event(); ev.record(main); s2.wait_event(ev)
        for _ in range(reps):
            for i in range(len(LAYERS)):
                if mode in ("serial", "dgrad"): dgrad(i, main)
                if mode in ("serial", "wgrad"): wgrad(i, main)
                if mode == "two": dgrad(i, main); wgrad(i, s2)
        if mode == "two":
            e2 = torch.cuda.Event(); e2.record(s2); main.wait_event(e2)
    body(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main); body(); e1.record(main); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for m in ("dgrad", "wgrad", "serial", "two", "serial", "two"):
    print("%-7s %.3f ms per pass over %d layers" % (m, run(m), len(LAYERS)), flush=True)
