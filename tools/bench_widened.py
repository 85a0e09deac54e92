#!/usr/bin/env python3
"""Measurement of the rows SURVEY 8(f) marked "next" (DESIGN.md section 7), each with the oracle's CPU time beside it:
  f1  colour-frame remap with smoothed maps   stabnet_warp_rev_bundle2      deploy_bundle.py:136-146       HBM-bound
  f2  history ring: stack assembly            (inside stabnet_deploy_frame; measured by bench.py's frame table)
  f3  training-sample assembly                stabnet_augment_pairs         get_data_mini_after.py:14-147  HBM-bound
  f4  max-inscribed-rectangle crop search     stabnet_crop_search           deploy_bundle.py:344-366       latency / integer
One JSON object on stdout: per row the GPU time (HIP events around `reps` back-to-back calls, inputs resident in HBM), the
ALGORITHMIC bytes and the achieved GB/s against the 8 TB/s HBM peak, and the oracle (kind "port", NumPy, one process) on a bounded
sample of the same work.
  python tools/bench_widened.py [--reps 50] [--no-cpu]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import stabnet_oracle as O
from stabnet_amd import data, synthetic, warp
from stabnet_amd.config import Config

PEAK_HBM_GBPS = 8000.0
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--no-cpu", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)


def gpu_time(fn, reps, graph=True):
    """us per call.  graph=True: `reps` calls captured into ONE hipGraph and replayed (the calls are 5-20 us kernels behind ~15 us of
    Python: eager back-to-back calls would time the host); falls back to eager calls when the capture fails."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if graph:
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(reps):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps
        except Exception as e:                                  # noqa: BLE001
            print("bench_widened: graph capture failed (%s), timing eager calls" % e, file=sys.stderr)
            torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps            # us per call


def cpu_time(fn, budget_s=8.0):
    t0 = time.perf_counter()
    n = 0
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 20:
            return el / n * 1e6, n


out = {"device": torch.cuda.get_device_name(0), "reps": a.reps, "rows": {}}

# ---- f1: warpRevBundle2 (maps shrunk by 4 and blown up again, pixel coordinates, fixed-point bilinear remap of the BGR frame)
for (H, W) in ((720, 1280), (1080, 1920)):
    ocfg = O.Config(height=H, width=W)
    rng = np.random.default_rng(H)
    theta = (rng.standard_normal((1, 50)) * 0.06).astype(np.float32)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    _, pts2 = O.get_4_pts(theta, ocfg)
    x_map, y_map, _ = O.maps_from_Hs(O.get_Hs(pts2, ocfg), H, W, ocfg)
    img_d, xm_d, ym_d = torch.from_numpy(img).to(dev), torch.from_numpy(x_map).to(dev), torch.from_numpy(y_map).to(dev)
    us = gpu_time(lambda: warp.warpRevBundle2(img_d, xm_d, ym_d), a.reps)
    hw = H * W
    # maps read once (8 HW), small maps written + re-read (2 * 8 HW / 16), frame gathered (3 HW), frame written (3 HW)
    bytes_alg = 8.0 * hw + 2 * 8.0 * hw / 16 + 3.0 * hw + 3.0 * hw
    row = {"what": "stabnet_warp_rev_bundle2 (map_shrink_kernel + remap_color_kernel), uint8 BGR %dx%d" % (W, H), "gpu_us": us,
           "algorithmic_bytes": bytes_alg, "achieved_gbps": bytes_alg / us / 1e3, "frac_of_8tbps": bytes_alg / us / 1e3 / PEAK_HBM_GBPS,
           "frames_per_s": 1e6 / us}
    if not a.no_cpu:
        cus, n = cpu_time(lambda: O.warpRevBundle2(img, x_map[0], y_map[0]))
        row["cpu_baseline"] = {"us": cus, "kind": "port", "cores": 1, "sample": "%d call(s) of oracle.warpRevBundle2 (NumPy)" % n, "gpu_over_cpu": cus / us}
    out["rows"]["f1_remap_%dp" % H] = row

# ---- f3: training-sample assembly, 8 pairs at 288x512 (BASELINE configs[2] batch)
N, H, W = 8, 288, 512
cfg, ocfg = Config(height=H, width=W, batch_size=N), O.Config(height=H, width=W)
raw = synthetic.make_raw_pairs(cfg, N, H, W, 77)
para, jitter, Hs = data.draw(np.random.default_rng(5), cfg, N, H, W)
t = lambda k: torch.from_numpy(raw[k]).to(dev)
st, un, fl, m1, m2 = t("stable"), t("unstable"), t("flow"), t("matches1"), t("matches2")
para_d, jit_d, Hs_d = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (para, jitter, Hs))
us = gpu_time(lambda: data.augment_pairs(st, un, fl, m1, raw["n1"], m2, raw["n2"], para_d, jit_d, Hs_d, cfg), max(5, a.reps // 5))
bc = cfg.before_ch
C_in = 2 * (bc + 1) + 2
# every input channel read (through the 1/0.9 up-scale: ~0.81 of its pixels) and every output channel written once; flow in + out
bytes_alg = 4.0 * N * H * W * (C_in + 2 * (2 * bc + 1) + 2 + 2 + 2)
row = {"what": "stabnet_augment_pairs, %d pairs at %dx%d (4 launches)" % (N, W, H), "gpu_us": us, "algorithmic_bytes": bytes_alg,
       "achieved_gbps": bytes_alg / us / 1e3, "frac_of_8tbps": bytes_alg / us / 1e3 / PEAK_HBM_GBPS, "pairs_per_s": N * 1e6 / us}
if not a.no_cpu:
    def one_pair():
        p = {"h": int(para[0, 0]), "w": int(para[0, 1]), "flip": int(para[0, 2])}
        O.assemble_pair(raw["stable"][0], raw["unstable"][0], raw["flow"][0], raw["matches1"][0], int(raw["n1"][0]), raw["matches2"][0],
                        int(raw["n2"][0]), p, jitter[0, 0], jitter[0, 1], Hs[0, 0].reshape(bc, 3, 3), Hs[0, 1].reshape(bc, 3, 3), ocfg)
    cus, n = cpu_time(one_pair)
    row["cpu_baseline"] = {"us_per_pair": cus, "kind": "port", "cores": 1, "sample": "%d pair(s) through oracle.assemble_pair (NumPy)" % n,
                           "gpu_over_cpu": cus * N / us}
out["rows"]["f3_augment_8x288x512"] = row

# ---- f4: crop search on the accumulated black mask of a clip
for (H, W) in ((720, 1280),):
    rng = np.random.default_rng(3)
    ab = np.zeros((H, W), np.int32)
    ab[: H // 12] = 5; ab[-H // 10:] = 3; ab[:, : W // 14] = 2; ab[:, -W // 16:] = 7       # borders that were black in some frame
    ab[rng.integers(0, H, 40), rng.integers(0, W, 40)] += 1
    ab_d = torch.from_numpy(ab).to(dev)
    us = gpu_time(lambda: warp.max_inscribed_rect(ab_d), max(5, a.reps // 5), graph=False)      # (reads its answer back: not capturable)
    row = {"what": "stabnet_crop_search %dx%d, step 10 (+ the 20-byte read-back)" % (W, H), "gpu_us": us, "answer": warp.max_inscribed_rect(ab_d)}
    if not a.no_cpu:
        cus, n = cpu_time(lambda: O.max_inscribed_rect(ab), budget_s=10.0)
        row["cpu_baseline"] = {"us": cus, "kind": "port", "cores": 1, "sample": "%d call(s) of oracle.max_inscribed_rect (vectorised NumPy; the reference's literal quadruple Python loop is ~100x slower)" % n,
                               "gpu_over_cpu": cus / us}
    out["rows"]["f4_crop_%dp" % H] = row

print(json.dumps(out))
