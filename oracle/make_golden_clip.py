#!/usr/bin/env python3
"""Generates tests/golden/clip_256x256_t64.npz: the SURVEY 8d Config-1 clip (BASELINE.json configs[0]: 256x256, 64
frames, seed 1234, seeded weights seed 0) run through the ORACLE's restatement of the deploy loop
(deploy_bundle.py:216-232,259-296,319-332), with per-frame checksums of x_map, y_map, black, out.

PARITY UNPINNED: these are outputs of the oracle, not of the reference (TensorFlow 1.3 cannot run here);
they pin the oracle against regressions (CPU test) and give the GPU stream test a fixed 63-step expected trajectory.

Per frame t = 1..63:  theta [50];  crc32 of the float32 bytes of x_map, y_map, black, out (exact: oracle regression);
float64 sums of the four tensors; black count; 16x16 strided samples of x_map, y_map, out (rows/cols ::16).
  python oracle/make_golden_clip.py            (about 1 minute on 8 cores)"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import stabnet_oracle as O  # noqa: E402

H = W = 256
T = 64
CLIP_SEED, WEIGHT_SEED, THETA_SCALE, STRIDE = 1234, 0, 0.2, 16


def frame_record(r):
    xm, ym = r["x_map"][0, :, :, 0], r["y_map"][0, :, :, 0]
    bl, out = r["black_pix"][0].astype(np.float32), r["output"][0, :, :, 0]
    crc = [zlib.crc32(np.ascontiguousarray(a, np.float32).tobytes()) for a in (xm, ym, bl, out)]
    sums = [float(np.sum(a, dtype=np.float64)) for a in (xm, ym, bl, out)]
    s = (slice(0, H, STRIDE), slice(0, W, STRIDE))
    return crc, sums, xm[s].copy(), ym[s].copy(), out[s].copy()


def run(n_frames=T):
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    cfg = Config(height=H, width=W)
    ocfg = O.Config(height=H, width=W)
    P = synthetic.make_params(cfg, seed=WEIGHT_SEED, theta_scale=THETA_SCALE)
    clip = synthetic.make_clip(H, W, T, seed=CLIP_SEED, margin=64)
    ring = O.DeployRing(clip[0], ocfg)
    rec = {k: [] for k in ("theta", "crc", "sums", "xs", "ys", "os")}
    for t in range(1, n_frames):
        r, _ = O.deploy_step(ring, clip[t], P, ocfg)
        crc, sums, xs, ys, os_ = frame_record(r)
        rec["theta"].append(r["theta"][0]); rec["crc"].append(crc); rec["sums"].append(sums)
        rec["xs"].append(xs); rec["ys"].append(ys); rec["os"].append(os_)
    return {"theta": np.stack(rec["theta"]).astype(np.float32), "crc": np.array(rec["crc"], np.uint32),
            "sums": np.array(rec["sums"], np.float64), "x_map_s": np.stack(rec["xs"]), "y_map_s": np.stack(rec["ys"]),
            "out_s": np.stack(rec["os"]),
            "meta": np.array([H, W, T, CLIP_SEED, WEIGHT_SEED, STRIDE], np.int64), "theta_scale": np.float64(THETA_SCALE)}


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden", "clip_256x256_t64.npz")
    np.savez_compressed(out, **run())
    print("wrote", out, os.path.getsize(out), "bytes")
