"""CPU tests of the oracle itself: known-answer tests derived from the reference source, the committed
golden vectors, and agreement between the NumPy forward oracle and the torch float64 gradient oracle."""
import glob
import os

import numpy as np
import pytest

from oracle import stabnet_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
F = np.float32


def test_inverse_restatement_matches_lapack():
    rng = np.random.default_rng(0)
    A = rng.standard_normal((500, 8, 8)).astype(F)
    X = O.inv8_partial_piv_lu(A)
    ref = np.linalg.inv(A.astype(np.float64))
    assert np.abs(X - ref).max() / np.abs(ref).max() < 5e-4
    assert np.abs(np.einsum('bij,bjk->bik', A.astype(np.float64), X.astype(np.float64)) - np.eye(8)).max() < 5e-3


def test_kat_identity_mesh():
    """(1) theta=0 -> regular grid, Hs = I +- ridge (1.0e-3), x_map ~ linspace; (9) pts1 packing."""
    cfg = O.Config(height=64, width=128)
    pts1, pts2 = O.get_4_pts(np.zeros((1, 50), F), cfg)
    gx, gy = np.meshgrid(np.linspace(-1, 1, 5), np.linspace(-1, 1, 5))
    assert np.array_equal(pts2[0, :, :, 0], gx.astype(F)) and np.array_equal(pts2[0, :, :, 1], gy.astype(F))
    assert np.array_equal(pts1[0, 1, 2], np.array([0, .5, 0, .5, -.5, -.5, 0, 0], F))   # xTL,xTR,xBL,xBR,yTL,yTR,yBL,yBR
    Hs = O.get_Hs(pts2, cfg)
    d = np.abs(Hs.reshape(16, 9) - np.eye(3, dtype=F).reshape(9)).max()
    assert 5e-4 < d < 1.2e-3                               # the ridge is semantic: max|H-I| = 1.0e-3
    x_map, y_map, black = O.maps_from_Hs(Hs, 64, 128, cfg)
    assert np.abs(x_map[0] - np.linspace(-1, 1, 128)[None]).max() < 2e-3
    assert black[0, 1:-1, 1:-1].sum() == 0


def test_kat_sampler_not_identity_and_border_zero():
    """(1),(8): pixel j samples j*W/(W-1); clipped corners make the weights cancel -> last column ~ 0."""
    H, W = 8, 16
    im = np.tile(np.arange(W, dtype=F)[None, :, None], (H, 1, 1))[None]
    xs = O.linspace_tf(-1, 1, W)
    ys = O.linspace_tf(-1, 1, H)
    x = np.tile(xs[None, :], (H, 1))[None, :, :, None]
    y = np.tile(ys[:, None], (1, W))[None, :, :, None]
    out = O.interpolate(im, x, y)[0, :, :, 0]
    want = np.arange(W) * W / (W - 1)
    assert np.abs(out[:-1, :-1] - want[None, :-1]).max() < 1e-4
    assert np.abs(out[:, -1]).max() < 1e-4 and np.abs(out[-1, :]).max() < 1e-4
    far = O.interpolate(im, np.full_like(x, 3.0), y)
    assert np.abs(far).max() < 1e-4                         # "~0": the weights cancel up to rounding


def test_kat_black_is_strict():
    """(2) exactly +-1 is NOT black (spatial_transformer3.py:284)."""
    cfg = O.Config(height=4, width=4)
    Hs = np.tile(np.eye(3, dtype=F).reshape(1, 1, 1, 9), (1, 4, 4, 1))
    x_map, y_map, black = O.maps_from_Hs(Hs, 4, 4, cfg)
    assert x_map[0, 0, 0] == -1.0 and x_map[0, 0, -1] <= 1.0
    assert black.sum() == 0


def test_kat_uniform_translation_and_seams():
    """(3) uniform vertex translation -> the same translation in every cell; (7) C0 seams up to the ridge error."""
    cfg = O.Config(height=64, width=128)
    theta = np.tile(np.array([0.05, -0.03], F), 25)[None]
    _, pts2 = O.get_4_pts(theta, cfg)
    x_map, y_map, _ = O.maps_from_Hs(O.get_Hs(pts2, cfg), 64, 128, cfg)
    assert np.abs(x_map[0] - (np.linspace(-1, 1, 128)[None] + 0.05)).max() < 3e-3
    assert np.abs(y_map[0] - (np.linspace(-1, 1, 64)[:, None] - 0.03)).max() < 3e-3
    rng = np.random.default_rng(3)
    _, pts2 = O.get_4_pts((rng.standard_normal((1, 50)) * 0.05).astype(F), cfg)
    x_map, y_map, _ = O.maps_from_Hs(O.get_Hs(pts2, cfg), 64, 128, cfg)
    step = np.abs(np.diff(x_map[0], axis=1))
    assert step[:, 31].max() < 2.5 * np.median(step)       # seam between cell columns 0|1 at x = 32


def test_kat_clip_and_dead_black_loss():
    """(4) vertices saturate at +-1.25; (5) black_pos loss == 0 because get_4_pts already clips."""
    cfg = O.Config()
    theta = np.full((2, 50), 3.0, F)
    pts1, pts2 = O.get_4_pts(theta, cfg)
    assert pts2.max() == 1.25 and O.get_black_pos(pts1, cfg).max() == 0
    pts1, _ = O.get_4_pts(-theta, cfg)
    assert O.get_black_pos(pts1, cfg).max() == 0


def test_kat_distortion_and_consistency_zero_sets():
    """(6) distortion = 0 for similarity meshes (k = 1), consistency = 0 for affine meshes."""
    cfg = O.Config()
    gx, gy = np.meshgrid(np.linspace(-1, 1, 5), np.linspace(-1, 1, 5))
    c, s = 0.9 * np.cos(0.1), 0.9 * np.sin(0.1)
    sim = np.stack([c * gx - s * gy + 0.02, s * gx + c * gy - 0.01], axis=-1)[None].astype(F)
    pts1 = np.stack([sim[:, :-1, :-1], sim[:, :-1, 1:], sim[:, 1:, :-1], sim[:, 1:, 1:]], axis=-1).reshape(1, 4, 4, 8)
    assert O.get_distortion_loss(pts1, cfg) < 1e-12
    aff = np.stack([1.1 * gx + 0.2 * gy, -0.1 * gx + 0.8 * gy + 0.05], axis=-1)[None].astype(F)
    assert O.get_consistency_loss(aff, cfg) < 1e-12
    assert O.get_consistency_loss(sim + np.random.default_rng(0).normal(0, .05, sim.shape).astype(F), cfg) > 1e-5


def test_warp_pts_rounds_half_to_even():
    cfg = O.Config(height=4, width=8)
    flow = np.arange(4 * 8 * 2, dtype=F).reshape(1, 4, 8, 2)
    pts = np.array([[[-1 + 2 * 2.5 / 8, -1.0], [-1 + 2 * 3.5 / 8, -1.0]]], F)   # x pixel 2.5 -> 2, 3.5 -> 4
    _, (xi, yi) = O.warp_pts(pts, flow, cfg)
    assert xi.tolist() == [[2, 4]] and yi.tolist() == [[0, 0]]


def test_deploy_ring_lags_and_channel_order():
    cfg = O.Config(height=2, width=2)
    ring = O.DeployRing(np.zeros((2, 2), F), cfg)
    assert len(ring.frames) == 32
    for t in range(1, 40):
        ring.push(np.full((2, 2), t, F), np.full((2, 2), -t, F))
    st = ring.stack(np.full((2, 2), 99, F))
    assert st.shape == (1, 2, 2, 13)
    assert st[0, 0, 0].tolist() == [-39, -38, -36, -32, -24, -8, 39, 38, 36, 32, 24, 8, 99]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "warp_*.npz"))))
def test_oracle_reproduces_golden_warp(path):
    g = np.load(path)
    gh, gw = g["grid"]
    N, H, W, C = g["U"].shape
    cfg = O.Config(height=H, width=W, grid_h=int(gh), grid_w=int(gw))
    pts1, pts2 = O.get_4_pts(g["theta"], cfg)
    out, black, img, Hs, _ = O.transformer(g["U"], pts2, cfg, return_all=True)
    for k, v in (("pts1", pts1), ("pts2", pts2), ("Hs", Hs), ("x_map", img[..., 0]), ("y_map", img[..., 1]), ("out", out)):
        assert np.array_equal(g[k], v), k
    assert np.array_equal(g["black"], black.astype(np.uint8))
    assert np.array_equal(g["interp"], O.interpolate(out, g["fx"], g["fy"]))


def test_oracle_reproduces_golden_losses():
    g = np.load(os.path.join(GOLD, "losses_32x64.npz"))
    N, H, W, _ = g["U"].shape
    cfg = O.Config(height=H, width=W, batch_size=N, max_matches=64)
    pts1, pts2 = O.get_4_pts(g["theta"], cfg)
    out, black, img = O.transformer(g["U"], pts2, cfg)
    feat, warped = O.feature_loss(g["matches"], g["mask"], img, cfg)
    assert np.isclose(g["distortion"], O.get_distortion_loss(pts1, cfg), rtol=1e-6)
    assert np.isclose(g["consistency"], O.get_consistency_loss(pts2, cfg), rtol=1e-6)
    assert np.isclose(g["feature"], feat, rtol=1e-6) and np.array_equal(g["warped"], warped)
    assert np.isclose(g["img_loss"], O.img_loss(out, g["y"], black, cfg), rtol=1e-6)


def test_torch_gradient_oracle_agrees_with_numpy_forward():
    import torch
    from oracle import torch_ref as T
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    N, H, W = 2, 32, 64
    cfg = O.Config(height=H, width=W, batch_size=N, max_matches=40)
    pcfg = Config(height=H, width=W, batch_size=N, max_matches=40)
    P = synthetic.make_params(pcfg, 0, 0.3)
    b = synthetic.make_train_batch(pcfg, N, H, W, 5)
    b["flow"] = b["flow"] + np.random.default_rng(1).normal(0, 0.03, b["flow"].shape).astype(F)
    # forward in moving-average mode (the NumPy oracle's batch-stat BN is the same formula; checked via training=True too)
    for training in (False, True):
        r = O.inference_stable_net(b["x1"], P, cfg, b["y1"], b["matches1"], b["mask1"], training=training)
        pt = {k: T.t(v) for k, v in P.items()}
        theta, id1, id2 = T.get_resnet(T.t(b["x1"]), pt, cfg, training)
        L = T.tower_losses(theta, T.t(b["x1"])[..., 12:13], T.t(b["y1"]), T.t(b["matches1"]), T.t(b["mask1"]), cfg)
        total = T.tower_total(id1, id2, L, T.regu_loss(pt, cfg), cfg)
        assert np.abs(theta.numpy() - r["theta"]).max() < 2e-5
        assert np.abs(L["flow"].numpy()[..., 0:1] - r["x_map"]).max() < 2e-4
        assert abs(float(L["distortion"]) - float(r["distortion_loss"])) < 1e-5
        assert abs(float(L["consistency"]) * cfg.consistency_mul - float(r["consistency_loss"])) < 1e-5
        assert abs(float(L["feature"]) - float(r["feature_loss"])) < 2e-3
        assert abs(float(L["img"]) * cfg.img_mul - float(r["img_loss"])) < 2e-2 * max(1.0, float(r["img_loss"]))
        assert abs(float(total) - float(r["total_loss"])) < 2e-2 * max(1.0, abs(float(r["total_loss"])))
    # temporal loss
    r1 = O.inference_stable_net(b["x1"], P, cfg)
    r2 = O.inference_stable_net(b["x2"], P, cfg)
    tn = O.temporal_loss(r1["output"], r1["black_pix"], r2["output"], r2["black_pix"], b["flow"], cfg)
    tt = T.temporal_loss(T.t(r1["output"]), T.t(r1["black_pix"]), T.t(r2["output"]), T.t(r2["black_pix"]), T.t(b["flow"]), cfg)
    assert abs(float(tt) - float(tn)) < 1e-4 * max(1.0, float(tn))


def test_cv_restatement_known_answers():
    """CPU KAT of the restatement: maps that address pixel centres exactly (x = j) return the frame itself."""
    H, W = 16, 24
    img = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
    mx = np.tile(np.arange(W, dtype=np.float32)[None], (H, 1))
    my = np.tile(np.arange(H, dtype=np.float32)[:, None], (1, W))
    assert np.array_equal(O.cv_remap_linear_u8(img, mx, my), img)
    half = O.cv_remap_linear_u8(img, mx + 0.5, my)                       # half-pixel shift = mean of neighbours ...
    a, b = img[:, :-1].astype(np.int64), img[:, 1:].astype(np.int64)
    assert np.array_equal(half[:, :-1], ((a + b + 1) >> 1).astype(np.uint8))   # ... rounded half UP: (16384 a + 16384 b + 16384) >> 15
    assert np.array_equal(half[:, -1], (img[:, -1].astype(np.int64) + 1) >> 1)  # the right tap of the last column is the border value 0
    # OpenCV's fixed-point weight table (initInterTab2D): sums to 32768 everywhere, exact multiples of 32 except entry (0, 0), whose
    # 1.0 saturates to 32767 and whose repair lands on tap [1][1]
    tab = O.cv_bilinear_tab_i()
    assert tab.shape == (1024, 4) and (tab.sum(1) == 32768).all() and tab.min() >= 0
    assert tab[0].tolist() == [32767, 0, 0, 1] and tab[1].tolist() == [31744, 1024, 0, 0] and tab[32 * 16 + 16].tolist() == [8192] * 4
    assert (tab[1:] % 32 == 0).all()
    # a quarter-pixel shift down-right: weights 9/16, 3/16, 3/16, 1/16 exactly
    q = O.cv_remap_linear_u8(img, mx + 0.25, my + 0.25)[:-1, :-1].astype(np.int64)
    p00, p01, p10, p11 = (img[:-1, :-1].astype(np.int64), img[:-1, 1:].astype(np.int64), img[1:, :-1].astype(np.int64), img[1:, 1:].astype(np.int64))
    assert np.array_equal(q, (18432 * p00 + 6144 * p01 + 6144 * p10 + 2048 * p11 + 16384) >> 15)
    # out-of-frame and non-finite coordinates read the border
    far = O.cv_remap_linear_u8(img, np.full_like(mx, 1e30), np.full_like(my, np.nan))
    assert not far.any()
    # resize: a constant map stays constant, a shrink by 4 of a ramp samples its centre of mass
    assert np.all(O.cv_resize_linear_f32(np.full((8, 8), 3.0, np.float32), 2, 2) == 3.0)
    ramp = np.tile(np.arange(8, dtype=np.float32)[None], (8, 1))
    assert np.allclose(O.cv_resize_linear_f32(ramp, 2, 2)[0], [1.5, 5.5])


def test_crop_search_restatement_equals_the_literal_reference_loop():
    """oracle.max_inscribed_rect vectorises the inner `ww` loop of deploy_bundle.py:344-366; check it against the literal
    quadruple loop (incl. first-found-wins ties) on small masks."""
    import math

    def literal(all_black, step):
        height, width = all_black.shape
        black_sum = np.zeros([height + 1, width + 1], dtype=np.int64)
        for i in range(height):
            for j in range(width):
                black_sum[i + 1][j + 1] = black_sum[i][j + 1] + black_sum[i + 1][j] - black_sum[i][j] + all_black[i][j]
        max_s, ans = 0, []
        for i in range(0, int(math.floor(height * 0.5)), step):
            for j in range(0, int(math.floor(width * 0.5)), step):
                if all_black[i][j] > 0:
                    continue
                for hh in range(i, height):
                    for ww in range(j, width):
                        if black_sum[hh + 1][ww + 1] - black_sum[hh + 1][j] - black_sum[i][ww + 1] + black_sum[i][j] > 0:
                            break
                        s = (hh - i + 1) * (ww - j + 1)
                        if s > max_s:
                            max_s, ans = s, [i, j, hh, ww]
        return ans, max_s

    rng = np.random.default_rng(1)
    for t, dens in enumerate([0.0, 0.01, 0.03, 0.1, 0.3, 1.0]):
        H, W = int(rng.integers(20, 44)), int(rng.integers(20, 50))
        m = (rng.random((H, W)) < dens).astype(np.int64)
        for step in (3, 10):
            assert literal(m, step) == O.max_inscribed_rect(m, step), (t, step)


def test_sample_assembly_known_answers():
    """KATs of the sample-assembly restatement (get_data_mini_after.py): resize at scale 1 is the identity; the identity
    homography blacks nothing and a translation by +2 in x blacks everything; contrast 1 / brightness 0 / no flip / zero
    crop offsets reproduce the top-left crop of the up-scaled frame; flipping twice is the identity on points."""
    rng = np.random.default_rng(3)
    img = rng.uniform(-0.5, 0.5, (18, 32)).astype(np.float32)
    assert np.array_equal(O.tf_resize_bilinear(img, 18, 32), img)
    up = O.tf_resize_bilinear(img, 36, 64)
    assert np.array_equal(up[::2, ::2], img)                                   # scale 1/2: even samples hit source pixels
    eye = np.eye(3, dtype=np.float32)
    assert O.rand_mask_from_H(eye, 18, 32).sum() == 0
    shift = eye.copy(); shift[0, 2] = 2.5
    assert O.rand_mask_from_H(shift, 18, 32).sum() == 18 * 32
    h, w = O.aug_resized_hw(18, 32)
    out = O.warp_img(img, {"h": 0, "w": 0, "flip": 0}, 1.0, 0.0)
    ref = O.tf_resize_bilinear(img, h, w)[:18, :32]
    assert np.abs(out - np.clip(ref, -0.5, 0.5)).max() <= 6e-8                 # (x-mean)*1+mean rounds once more
    pts = rng.uniform(-0.8, 0.8, (5, 4)).astype(np.float32)
    p0, _ = O.warp_point(pts, np.ones(5, bool), {"h": 0, "w": 0, "flip": 0}, 18, 32)
    p1, _ = O.warp_point(pts, np.ones(5, bool), {"h": 0, "w": 0, "flip": 1}, 18, 32)
    assert np.allclose(p1[:, [0, 2]], -p0[:, [0, 2]] - 1.0 / 32, atol=1e-6) and np.array_equal(p1[:, [1, 3]], p0[:, [1, 3]])
