#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the oracle itself.

PARITY UNPINNED: the reference cannot run here and ships no fixtures, so these vectors are not
reference outputs; they freeze the oracle's behaviour at the commit that made them (regression guard) and give the
GPU tests a fixed input/expected-output set that does not depend on importing the oracle's generator code paths.
  python oracle/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import stabnet_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def warp_case(name, N, H, W, C, gh, gw, std, seed):
    cfg = O.Config(height=H, width=W, grid_h=gh, grid_w=gw)
    rng = np.random.default_rng(seed)
    theta = (rng.standard_normal((N, (gh + 1) * (gw + 1) * 2)) * std).astype(np.float32)
    U = (rng.random((N, H, W, C)) - 0.5).astype(np.float32)
    pts1, pts2 = O.get_4_pts(theta, cfg)
    out, black, img, Hs, _ = O.transformer(U, pts2, cfg, return_all=True)
    fx = (img[..., 0:1] + rng.normal(0, 0.02, (N, H, W, 1))).astype(np.float32)
    fy = (img[..., 1:2] + rng.normal(0, 0.02, (N, H, W, 1))).astype(np.float32)
    interp = O.interpolate(out, fx, fy)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), theta=theta, U=U, pts1=pts1, pts2=pts2, Hs=Hs,
                        x_map=img[..., 0], y_map=img[..., 1], black=black.astype(np.uint8), out=out, fx=fx, fy=fy,
                        interp=interp, grid=np.array([gh, gw]))


def losses_case(name, N, H, W, seed):
    cfg = O.Config(height=H, width=W, batch_size=N, max_matches=64)
    rng = np.random.default_rng(seed)
    theta = (rng.standard_normal((N, 50)) * 0.08).astype(np.float32)
    U = (rng.random((N, H, W, 1)) - 0.5).astype(np.float32)
    y = (rng.random((N, H, W, 1)) - 0.5).astype(np.float32)
    matches = rng.uniform(-1, 1, (N, 64, 4)).astype(np.float32)
    mask = (rng.random((N, 64)) < 0.5).astype(np.float32)
    pts1, pts2 = O.get_4_pts(theta, cfg)
    out, black, img = O.transformer(U, pts2, cfg)
    feat, warped = O.feature_loss(matches, mask, img, cfg)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), theta=theta, U=U, y=y, matches=matches, mask=mask,
                        distortion=O.get_distortion_loss(pts1, cfg), consistency=O.get_consistency_loss(pts2, cfg),
                        feature=feat, warped=warped, img_loss=O.img_loss(out, y, black, cfg),
                        black_pos=O.get_black_pos(pts1, cfg))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    warp_case("warp_32x64", 2, 32, 64, 1, 4, 4, 0.05, 11)
    warp_case("warp_45x77", 2, 45, 77, 1, 4, 4, 0.08, 12)      # remainder row/col, W % 4 != 0
    warp_case("warp_48x40_c3_g2x3", 1, 48, 40, 3, 2, 3, 0.1, 13)
    warp_case("warp_clip_36x52", 1, 36, 52, 1, 4, 4, 0.7, 14)  # saturating vertices / folded cells
    losses_case("losses_32x64", 2, 32, 64, 21)
    print("wrote", sorted(os.listdir(OUT)))
