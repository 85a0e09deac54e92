#!/usr/bin/env python3
"""Multi-GPU readiness on ONE GPU (VERDICT r3 #7b): can a collective's kernel run beside the training backward, whose persistent
convolution kernels fill every CU (3 workgroups x 48 KB of LDS), and what does the step pay?  A one-rank RCCL group moves nothing, so
the collective is played by stabnet_probe_comm_proxy: W long-lived 256-thread workgroups (RCCL runs one per channel) holding L bytes
of LDS that stream `dst += src` over a gradient-sized bucket (121.6 MB) on a SECOND stream, one launch per training step, enqueued when
the step is.  Reported per configuration: the step time with and without the proxy, the proxy's own duration beside the step and
alone, and how long after its launch the LAST of its workgroups got a CU slot (in-kernel s_memrealtime stamps).
`reserved` = stabnet_conv_reserve_cus(k): the persistent conv grids sized for 256 - k CUs.
  python tools/comm_proxy.py [--steps 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stabnet_amd import _lib, synthetic
from stabnet_amd.config import Config
from stabnet_amd.train import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
N, H, W = 8, 288, 512
cfg = Config(height=H, width=W, batch_size=N)
tr = Trainer(synthetic.make_params(cfg, seed=0, theta_scale=0.2), N, H, W, cfg, device=dev)
b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_train_batch(cfg, N, H, W, seed=1234).items()}
gates = {"use_theta_loss": 1, "use_temp_loss": 1, "use_black_loss": 1, "use_theta_only": 0}
L = _lib.lib()
nfl = int(tr.plan.n_trainable) // 4 * 4
src = torch.zeros(nfl, dtype=torch.float32, device=dev)
dst = torch.zeros(nfl, dtype=torch.float32, device=dev)
side = torch.cuda.Stream(device=dev)


def run(steps, wgs=0, lds=0):
    """-> (ms per step, proxy ms beside the step, last-workgroup start delay in us)"""
    stamps = torch.zeros(2 * max(wgs, 1), dtype=torch.int64, device=dev)
    for _ in range(3):
        tr.forward_backward(b, gates)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pe = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    delays = []
    e0.record()
    for i in range(steps):
        if wgs:
            side.wait_stream(torch.cuda.current_stream(dev))       # the proxy of step i starts when step i does
            with torch.cuda.stream(side):
                pe[i][0].record(side)
                _lib.call("stabnet_probe_comm_proxy", src.data_ptr(), dst.data_ptr(), nfl, wgs, lds, stamps.data_ptr() if i == steps - 1 else 0,
                          side.cuda_stream, device=dev)
                pe[i][1].record(side)
        tr.forward_backward(b, gates)
    e1.record()
    torch.cuda.synchronize()
    pms = float(np.median([x.elapsed_time(y) for x, y in pe])) if wgs else 0.0
    if wgs:
        s = stamps.view(-1, 2).cpu().numpy()
        delays = (s[:, 0] - s[:, 0].min()) / 100.0
    return e0.elapsed_time(e1) / steps, pms, (float(np.max(delays)) if wgs else 0.0)


def proxy_alone(wgs, lds, reps=5):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record(side)
        for _ in range(reps):
            _lib.call("stabnet_probe_comm_proxy", src.data_ptr(), dst.data_ptr(), nfl, wgs, lds, 0, side.cuda_stream, device=dev)
        e1.record(side)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


base, _, _ = run(a.steps)
print("training step alone: %.3f ms (%.1f pairs/s); bucket %.1f MB" % (base, 1e3 * N / base, nfl * 4 / 1e6))
print("%8s %5s %8s | %9s %8s | %10s %10s %12s" % ("reserved", "WGs", "LDS", "step ms", "vs alone", "proxy ms", "alone ms", "last WG +us"))
for reserved in (0, 16, 32):
    L.stabnet_conv_reserve_cus(reserved)
    alone_step, _, _ = run(a.steps)
    for wgs, lds in ((32, 16384), (64, 16384), (32, 49152)):
        pa = proxy_alone(wgs, lds)
        ms, pms, dl = run(a.steps, wgs, lds)
        print("%8d %5d %8d | %9.3f %+7.1f%% | %10.3f %10.3f %12.1f   (no proxy, this reservation: %.3f ms)" % (
            reserved, wgs, lds, ms, 100.0 * (ms / base - 1.0), pms, pa, dl, alone_step))
L.stabnet_conv_reserve_cus(0)
