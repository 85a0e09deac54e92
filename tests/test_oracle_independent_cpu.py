"""CPU: narrows the UNPINNED oracle (the reference ships no fixtures and cannot run here) with INDEPENDENT
implementations of its `[external]` pieces -- code written separately from oracle/stabnet_oracle.py (and from
oracle/torch_ref.py, which shares structure with it), on different libraries:

  backbone   torch.nn.functional conv2d / max_pool2d / batch_norm composed here from SURVEY Appendix A
  8x8 LU     scipy.linalg.lu pivot sequence + float64 inverse, on the ill-conditioned ridge systems of SURVEY section 7
  resizes    torch interpolate (half-pixel, cv2 convention), scipy.ndimage.map_coordinates (TF1 legacy convention)
  sampler    torch grid_sample with the half-pixel shift, interior samples
  Adam       closed forms + float64 shadow
  clip       the committed 256x256 trajectory (tests/golden/clip_256x256_t64.npz)

None of this pins the oracle to TensorFlow's outputs; it shows two independent readings of the same published
algorithms agree."""
import os
import zlib

import numpy as np
import pytest
import scipy.linalg
import scipy.ndimage
import torch
import torch.nn.functional as TF

from oracle import stabnet_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_256x256_t64.npz")


# ----------------------------------------------------------------------------------------------- backbone
def _pad_same(x, k, s):
    """TF 'SAME': total = max((ceil(n/s)-1)*s + k - n, 0), before = total // 2, after = the rest (NCHW tensor)."""
    pads = []
    for n in (x.shape[3], x.shape[2]):                        # F.pad order: W first, then H
        out = -(-n // s)
        tot = max((out - 1) * s + k - n, 0)
        pads += [tot // 2, tot - tot // 2]
    return pads


def _independent_resnet_v2_50(x_nhwc, p, eps, training, dtype):
    """slim resnet_v2_50(global_pool=False, output_stride=32) + mean + FC head, from SURVEY Appendix A, in NCHW torch."""
    def w(name):                                              # HWIO -> OIHW
        return torch.from_numpy(np.asarray(p[name])).to(dtype).permute(3, 2, 0, 1).contiguous()

    def v(name):
        return torch.from_numpy(np.asarray(p[name])).to(dtype)

    def bn_relu(t, prefix):
        y = TF.batch_norm(t, None if training else v(prefix + "/moving_mean").clone(),
                          None if training else v(prefix + "/moving_variance").clone(), v(prefix + "/gamma"),
                          v(prefix + "/beta"), training=training, momentum=0.0, eps=eps)
        return torch.relu(y)

    def conv_same(t, name, k, s, bias=None):
        if s == 1:
            t = TF.pad(t, _pad_same(t, k, 1))
        else:                                                 # slim conv2d_same: explicit pad (k-1)//2 | rest, then VALID
            b = (k - 1) // 2
            t = TF.pad(t, [b, k - 1 - b, b, k - 1 - b])
        return TF.conv2d(t, w(name), bias, stride=s)

    R = "resnet_v2_50/"
    t = torch.from_numpy(np.asarray(x_nhwc)).to(dtype).permute(0, 3, 1, 2)
    t = conv_same(t, R + "conv1/weights", 7, 2, v(R + "conv1/biases"))
    t = TF.max_pool2d(TF.pad(t, _pad_same(t, 3, 2), value=float("-inf")), 3, 2)
    taps = {"pool1": t}
    for bname, depth, dbn, units, bstride in (("block1", 256, 64, 3, 2), ("block2", 512, 128, 4, 2),
                                              ("block3", 1024, 256, 6, 2), ("block4", 2048, 512, 3, 1)):
        for u in range(1, units + 1):
            s = bstride if u == units else 1
            S = R + "%s/unit_%d/bottleneck_v2/" % (bname, u)
            pre = bn_relu(t, S + "preact")
            if t.shape[1] == depth:
                sc = t if s == 1 else TF.max_pool2d(t, 1, s)
            else:
                sc = TF.conv2d(pre, w(S + "shortcut/weights"), v(S + "shortcut/biases"), stride=s)
            r = bn_relu(TF.conv2d(pre, w(S + "conv1/weights")), S + "conv1/BatchNorm")
            r = bn_relu(conv_same(r, S + "conv2/weights", 3, s), S + "conv2/BatchNorm")
            r = TF.conv2d(r, w(S + "conv3/weights"), v(S + "conv3/biases"))
            t = sc + r
            taps["%s/unit_%d" % (bname, u)] = t
    t = bn_relu(t, R + "postnorm")
    g = t.mean(dim=(2, 3))
    for k in (1, 2, 3):
        g = torch.relu(g @ v("fc/fc/fc_%d/weights" % k) + v("fc/fc/fc_%d/biases" % k))
    theta = g @ v("fc/fc_weights") + v("fc/fc_bias")
    return theta, taps


@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("H,W", [(64, 96), (90, 130)])        # 90x130: odd feature maps -> asymmetric SAME pads matter
def test_backbone_matches_independent_torch_functional(H, W, training):
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    cfg, ocfg = Config(height=H, width=W), O.Config(height=H, width=W)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
    x, _ = synthetic.make_stack(cfg, 2, H, W, seed=3)
    taps = {}
    theta, _, _ = O.get_resnet(x, P, ocfg, training=training, taps=taps)
    with torch.no_grad():
        th64, taps64 = _independent_resnet_v2_50(x, P, ocfg.bn_eps, training, torch.float64)
    for name in ("pool1", "block1/unit_3", "block2/unit_4", "block3/unit_6", "block4/unit_3"):
        want = taps64[name].permute(0, 2, 3, 1).numpy()
        assert taps[name].shape == want.shape, name
        assert np.abs(taps[name] - want).max() <= 3e-5 * np.abs(want).max(), name
    assert np.abs(theta - th64.numpy()).max() <= 1e-5


def test_max_pool_same_is_asymmetric():
    """Even input: TF-SAME pads (0 before, 1 after) -- NOT torch's symmetric padding=1."""
    x = np.random.default_rng(0).standard_normal((1, 8, 10, 3)).astype(np.float32)
    got = O.max_pool_3x3_s2_same(x)
    t = torch.from_numpy(x).permute(0, 3, 1, 2)
    want = TF.max_pool2d(TF.pad(t, [0, 1, 0, 1], value=float("-inf")), 3, 2).permute(0, 2, 3, 1).numpy()
    sym = TF.max_pool2d(t, 3, 2, padding=1).permute(0, 2, 3, 1).numpy()
    assert np.array_equal(got, want) and not np.array_equal(got, sym)


# ----------------------------------------------------------------------------------------------- 8x8 ridge inverse
def _ridge_systems(n, rng, std):
    """A + 1e-4 I of get_H for random meshes; cells touching a zero coordinate are the cond ~ 3e4 cases."""
    cfg = O.Config()
    theta = (rng.standard_normal((n, 50)) * std).astype(np.float32)
    _, pts2 = O.get_4_pts(theta, cfg)
    mats = []
    for i in range(4):
        for j in range(4):
            hh, ww = i * 0.5 - 1, j * 0.5 - 1
            ori = np.array([ww, hh, ww + 0.5, hh, ww, hh + 0.5, ww + 0.5, hh + 0.5], np.float32)
            tar = np.concatenate([pts2[:, i, j], pts2[:, i, j + 1], pts2[:, i + 1, j], pts2[:, i + 1, j + 1]], axis=1)
            x, y = ori[0::2], ori[1::2]
            u, v = tar[:, 0::2], tar[:, 1::2]
            A = np.zeros((n, 8, 8), np.float32)
            for r in range(4):
                A[:, r, 0:3] = (x[r], y[r], 1)
                A[:, r, 6], A[:, r, 7] = -x[r] * u[:, r], -y[r] * u[:, r]
                A[:, 4 + r, 3:6] = (x[r], y[r], 1)
                A[:, 4 + r, 6], A[:, 4 + r, 7] = -x[r] * v[:, r], -y[r] * v[:, r]
            mats.append(A + np.eye(8, dtype=np.float32) * np.float32(1e-4))
    return np.concatenate(mats)


def test_inv8_against_scipy_lu_and_float64_inverse():
    rng = np.random.default_rng(5)
    A = _ridge_systems(40, rng, 0.1)                            # 640 systems
    inv = O.inv8_partial_piv_lu(A)
    cond = np.linalg.cond(A.astype(np.float64))
    assert cond.max() > 1e4                                     # the ill-conditioned cells are in the sample
    worst = 0.0
    for k in range(len(A)):
        a64 = A[k].astype(np.float64)
        ref = np.linalg.inv(a64)
        # forward error of a backward-stable float32 LU: <~ cond * eps32 relative
        rel = np.abs(inv[k] - ref).max() / np.abs(ref).max()
        assert rel <= 40 * cond[k] * np.finfo(np.float32).eps, (k, rel, cond[k])
        worst = max(worst, np.abs(inv[k].astype(np.float64) @ a64 - np.eye(8)).max())
        # same pivot sequence as LAPACK's partial pivoting (scipy.linalg.lu) run in float32 on the same matrix,
        # whenever the pivot choice is not a near-tie (a tie may legitimately go either way)
        p_mat, l, u = scipy.linalg.lu(A[k])
        perm_scipy = np.argmax(p_mat, axis=0)                   # row of A placed at position i
        lu, perm = _lu_pivots_of_restatement(A[k])
        if not _has_near_tie(A[k]):
            assert np.array_equal(perm, perm_scipy), k
            assert np.abs(np.triu(lu) - u).max() <= 1e-4 * np.abs(u).max(), k
    assert worst < 5e-2


def _lu_pivots_of_restatement(a):
    """Pivot order the oracle's algorithm takes (first argmax of |column| below the diagonal), replayed on one matrix."""
    a = a.astype(np.float32).copy()
    perm = np.arange(8)
    for k in range(8):
        piv = int(np.argmax(np.abs(a[k:, k]))) + k
        a[[k, piv]] = a[[piv, k]]
        perm[[k, piv]] = perm[[piv, k]]
        a[k + 1:, k] = a[k + 1:, k] / a[k, k]
        a[k + 1:, k + 1:] = a[k + 1:, k + 1:] - np.outer(a[k + 1:, k], a[k, k + 1:])
    return a, perm


def _has_near_tie(a):
    a = a.astype(np.float64).copy()
    for k in range(8):
        col = np.abs(a[k:, k])
        order = np.sort(col)[::-1]
        if len(order) > 1 and order[1] > 0 and (order[0] - order[1]) <= 1e-5 * order[0]:
            return True
        piv = int(np.argmax(col)) + k
        a[[k, piv]] = a[[piv, k]]
        a[k + 1:, k] /= a[k, k]
        a[k + 1:, k + 1:] -= np.outer(a[k + 1:, k], a[k, k + 1:])
    return False


def test_maps_are_well_conditioned_even_when_H_is_not():
    """|dH| between float32 and float64 solves reaches 1e-2 on cond 3e4 cells while the induced map
    error stays at the 1e-6 level -- the reason the homography tolerance is stated on x_map / y_map."""
    rng = np.random.default_rng(9)
    cfg = O.Config(height=72, width=128)
    theta = (rng.standard_normal((64, 50)) * 0.1).astype(np.float32)
    _, pts2 = O.get_4_pts(theta, cfg)
    Hs32 = O.get_Hs(pts2, cfg)
    # float64 shadow of the same ridge solve
    Hs64 = np.empty(Hs32.shape, np.float64)
    for i in range(4):
        for j in range(4):
            hh, ww = i * 0.5 - 1, j * 0.5 - 1
            x = np.array([ww, ww + 0.5, ww, ww + 0.5]); y = np.array([hh, hh, hh + 0.5, hh + 0.5])
            tar = np.stack([pts2[:, i, j], pts2[:, i, j + 1], pts2[:, i + 1, j], pts2[:, i + 1, j + 1]], axis=1).astype(np.float64)
            for n in range(len(theta)):
                u, v = tar[n, :, 0], tar[n, :, 1]
                A = np.zeros((8, 8))
                A[:4, 0], A[:4, 1], A[:4, 2], A[:4, 6], A[:4, 7] = x, y, 1, -x * u, -y * u
                A[4:, 3], A[4:, 4], A[4:, 5], A[4:, 6], A[4:, 7] = x, y, 1, -x * v, -y * v
                h = np.linalg.solve(A + 1e-4 * np.eye(8), np.concatenate([u, v]))
                Hs64[n, i, j] = np.append(h, 1.0)
    x32, y32, _ = O.maps_from_Hs(Hs32, 72, 128, cfg)
    x64, y64, _ = O.maps_from_Hs(Hs64.astype(np.float32), 72, 128, cfg)
    # (compared where the map is near the frame: close to a homography's pole z -> 0 the map itself is unbounded)
    near = (np.abs(x64) < 2) & (np.abs(y64) < 2)
    assert near.mean() > 0.95
    dH = np.abs(Hs32 - Hs64).max()
    dmap = max(np.abs(x32 - x64)[near].max(), np.abs(y32 - y64)[near].max())
    assert dH > 100 * dmap and dmap < 1e-4, (dH, dmap)


# ----------------------------------------------------------------------------------------------- resizes / sampler
@pytest.mark.parametrize("src_hw,dst_hw", [((72, 128), (18, 32)), ((18, 32), (72, 128)), ((45, 77), (11, 19))])
def test_cv_resize_linear_matches_torch_half_pixel(src_hw, dst_hw):
    """cv2 INTER_LINEAR = half-pixel centres, no antialias = torch bilinear align_corners=False."""
    rng = np.random.default_rng(3)
    src = rng.standard_normal(src_hw).astype(np.float32)
    got = O.cv_resize_linear_f32(src, dst_hw[1], dst_hw[0])
    want = TF.interpolate(torch.from_numpy(src)[None, None].double(), size=dst_hw, mode="bilinear", align_corners=False,
                          antialias=False)[0, 0].numpy()
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-5


def test_tf_resize_bilinear_legacy_matches_map_coordinates():
    """TF1 ResizeBilinear(align_corners=False): source coordinate = dst * (in/out), no half-pixel offset."""
    rng = np.random.default_rng(4)
    img = rng.standard_normal((36, 64)).astype(np.float32)
    oh, ow = 40, 71
    got = O.tf_resize_bilinear(img, oh, ow)
    yy, xx = np.meshgrid(np.arange(oh) * (36 / oh), np.arange(ow) * (64 / ow), indexing="ij")
    want = scipy.ndimage.map_coordinates(img.astype(np.float64), [yy, xx], order=1, mode="nearest")
    # (TF forms the source coordinate in float32: at x ~ 60 its rounding is 4e-6 px, times the neighbour difference)
    assert np.abs(got - want).max() < 3e-5


def test_sampler_matches_grid_sample_on_interior_points():
    """spatial_transformer3.py:81-82 un-normalises with (x+1)*W/2 (no half-pixel offset): equals torch grid_sample
    (align_corners=False, which uses ((x+1)*W-1)/2) fed with x + 1/W, as long as all four corners are in frame."""
    rng = np.random.default_rng(6)
    N, H, W = 2, 24, 40
    im = rng.standard_normal((N, H, W, 1)).astype(np.float32)
    x = rng.uniform(-0.9, 0.85, (N, H, W, 1)).astype(np.float32)
    y = rng.uniform(-0.9, 0.85, (N, H, W, 1)).astype(np.float32)
    got = O.interpolate(im, x, y)
    grid = torch.from_numpy(np.concatenate([x + 1.0 / W, y + 1.0 / H], axis=3)).double()
    want = TF.grid_sample(torch.from_numpy(im).permute(0, 3, 1, 2).double(), grid, mode="bilinear", padding_mode="zeros",
                          align_corners=False).permute(0, 2, 3, 1).numpy()
    assert np.abs(got - want).max() < 1e-5


# ----------------------------------------------------------------------------------------------- optimiser
def test_adam_tf_closed_forms():
    a = O.AdamTF(3)
    g = np.array([1e-3, 1e-9, -1e-12], np.float32)
    w1 = a.step(np.zeros(3, np.float32), g, 2e-5)
    eps_hat = 1e-8 / np.sqrt(1e-3)                              # step 1: dw = -lr g / (|g| + eps / sqrt(1 - beta2))
    assert np.allclose(w1, -2e-5 * g.astype(np.float64) / (np.abs(g.astype(np.float64)) + eps_hat), rtol=1e-4, atol=0)
    # constant gradient: m_hat = g, v_hat = g^2 at every step -> every step moves by lr * g / (|g| + eps_t)
    a = O.AdamTF(1, dtype=np.float64)
    w = np.zeros(1)
    for t in range(1, 20):
        w_new = a.step(w, np.array([0.5]), 1e-3)
        assert (w - w_new)[0] == pytest.approx(1e-3, rel=1e-6), t
        w = w_new
    # float32 TF arithmetic stays within float32 rounding of the float64 shadow over 200 steps
    rng = np.random.default_rng(1)
    a32, a64 = O.AdamTF(500), O.AdamTF(500, dtype=np.float64)
    w32 = np.zeros(500, np.float32); w64 = np.zeros(500)
    for t in range(200):
        g = rng.standard_normal(500).astype(np.float32)
        w32 = a32.step(w32, g, 2e-5); w64 = a64.step(w64, g, 2e-5)
    assert np.abs(w32 - w64).max() < 1e-6 * 200 * 2e-5 / 2e-5 * 1e-3        # << one step (2e-5)
    assert float(O.exponential_decay_staircase(2e-5, 39999, 40000, 0.1)) == pytest.approx(2e-5)
    assert float(O.exponential_decay_staircase(2e-5, 40000, 40000, 0.1)) == pytest.approx(2e-6, rel=1e-6)
    assert float(O.exponential_decay_staircase(2e-5, 99999, 40000, 0.1)) == pytest.approx(2e-7, rel=1e-6)


# ----------------------------------------------------------------------------------------------- committed clip
def test_golden_clip_is_reproduced_by_the_oracle():
    """(1) the first frames of the free-running oracle loop reproduce the committed theta (BLAS summation order may
    differ between hosts: 1e-6; only the FIRST frames -- the loop's discontinuities, black mask and clipped-corner sampler,
    let two float32 runs drift apart later, see tests/test_baseline_sizes_gpu.py); (2) for EVERY frame the committed CRC32 checksums of x_map / y_map / black / out follow
    bit-exactly from the committed theta (the warp half of the oracle has no BLAS in it)."""
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    g = np.load(GOLDEN)
    H, W, Tn, clip_seed, weight_seed, stride = (int(v) for v in g["meta"])
    cfg, ocfg = Config(height=H, width=W), O.Config(height=H, width=W)
    P = synthetic.make_params(cfg, seed=weight_seed, theta_scale=float(g["theta_scale"]))
    clip = synthetic.make_clip(H, W, Tn, seed=clip_seed, margin=64)
    ring = O.DeployRing(clip[0], ocfg)
    for t in (1, 2, 3):
        r, _ = O.deploy_step(ring, clip[t], P, ocfg)
        assert np.abs(r["theta"][0] - g["theta"][t - 1]).max() < 1e-6, t
    assert g["theta"].shape == (Tn - 1, 50) and np.abs(g["theta"]).max() < 0.5
    for t in range(1, Tn):
        _, pts2 = O.get_4_pts(g["theta"][t - 1:t], ocfg)
        out, black, img = O.transformer(clip[t].reshape(1, H, W, 1), pts2, ocfg)
        arrs = (img[0, :, :, 0], img[0, :, :, 1], black[0].astype(np.float32), out[0, :, :, 0])
        crc = [zlib.crc32(np.ascontiguousarray(a, np.float32).tobytes()) for a in arrs]
        assert crc == [int(c) for c in g["crc"][t - 1]], t
        assert np.array_equal(arrs[0][::stride, ::stride], g["x_map_s"][t - 1])
