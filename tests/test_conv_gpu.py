"""GPU parity of the fp32-MFMA implicit-GEMM convolution (through the C ABI) against the oracle's im2col+sgemm
convolution; tolerance 2e-5 of the output scale (different summation order, both fp32-accumulate)."""
import numpy as np
import pytest
import torch

from oracle import stabnet_oracle as O

pytestmark = pytest.mark.gpu

# N,H,W,Cin,Cout,k,stride,pad, prologue, bias, residual(0 none,1 same,2 strided), relu
CASES = [
    (1, 20, 24, 16, 64, 7, 2, 3, False, True, 0, False),     # stem-like (padded 13->16 input), BK=16 path
    (1, 36, 64, 64, 64, 1, 1, 0, True, False, 0, False),     # bottleneck conv1
    (2, 36, 64, 64, 64, 3, 1, 1, True, False, 0, False),     # bottleneck conv2, SAME
    (1, 36, 64, 64, 256, 1, 1, 0, True, True, 1, False),     # conv3 + bias + residual
    (1, 37, 63, 128, 128, 3, 2, 1, True, False, 0, False),   # stride-2 conv2, odd sizes
    (1, 18, 32, 128, 512, 1, 1, 0, True, True, 2, False),    # conv3 + strided identity shortcut
    (1, 9, 16, 512, 512, 3, 1, 1, True, False, 0, True),     # small M, big K -> split-K
    (1, 9, 16, 2048, 512, 1, 1, 0, True, False, 0, False),   # block4 conv1 -> split-K
    (3, 72, 128, 64, 256, 1, 1, 0, False, True, 0, False),   # big M, no prologue -> LDS-DMA ring kernel, 2 K-steps
    # ring kernel (no prologue, Cin % 32 == 0): every K-step count of the steady loop / tail split, padding, stride
    (1, 17, 23, 32, 64, 1, 1, 0, False, False, 0, False),    # 1 K-step, ragged M
    (1, 17, 23, 96, 96, 1, 1, 0, False, True, 0, True),      # 3 K-steps, ragged Cout tile
    (1, 17, 23, 128, 64, 1, 1, 0, False, False, 1, False),   # 4 K-steps + residual
    (1, 17, 23, 160, 64, 1, 1, 0, False, False, 0, False),   # 5 K-steps (first steady iteration)
    (1, 17, 23, 256, 64, 1, 1, 0, False, False, 0, False),   # 8 K-steps
    (2, 36, 64, 64, 64, 3, 1, 1, False, False, 0, False),    # 3x3 SAME, zero page taps
    (1, 37, 63, 128, 128, 3, 2, 1, False, True, 0, True),    # 3x3 stride 2, odd sizes
    (1, 36, 64, 64, 64, 3, 1, 2, False, False, 0, False),    # pad 2 (full correlation, the stride-1 dgrad geometry)
    (1, 9, 16, 512, 512, 3, 1, 1, False, False, 0, True),    # split-K through the ring kernel
    (1, 18, 32, 128, 512, 1, 2, 0, False, True, 2, False),   # strided 1x1 (shortcut) + strided residual read
    (1, 180, 320, 64, 64, 3, 1, 1, False, True, 1, True),    # 900 tiles > 768 resident workgroups: several tiles per
                                                             # workgroup, the ring running across tile boundaries (MODE 1)
    (1, 180, 320, 32, 256, 1, 1, 0, False, False, 1, False), # 3600 one-step tiles: every step is a tile boundary (MODE 0)
]


@pytest.mark.parametrize("out_bn", [False, True])
@pytest.mark.parametrize("N,H,W,Cin,Cout,k,stride,pad,prologue,bias,res,relu", CASES)
def test_conv2d_matches_oracle(cuda, N, H, W, Cin, Cout, k, stride, pad, prologue, bias, res, relu, out_bn):
    from stabnet_amd import ops
    rng = np.random.default_rng(Cin * 7 + Cout + k)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32) if bias else None
    sc = rng.uniform(0.5, 1.5, Cin).astype(np.float32) if prologue else None
    sh = (rng.standard_normal(Cin) * 0.3).astype(np.float32) if prologue else None
    a = x if not prologue else np.maximum(x * sc + sh, 0).astype(np.float32)
    want = O.conv2d(a, w, stride, ((pad, pad), (pad, pad)), b)
    Ho, Wo = want.shape[1:3]
    r = None
    if res == 1:
        r = rng.standard_normal((N, Ho, Wo, Cout)).astype(np.float32)
        want = want + r
    elif res == 2:
        r = rng.standard_normal((N, 2 * Ho - 1, 2 * Wo, Cout)).astype(np.float32)
        want = want + r[:, ::2, ::2, :]
    osc = osh = None
    if out_bn:
        osc = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
        osh = (rng.standard_normal(Cout) * 0.3).astype(np.float32)
        want = (want * osc + osh).astype(np.float32)
    if relu:
        want = np.maximum(want, 0)
    t = lambda v: None if v is None else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(cuda)
    got = ops.conv2d(t(x), t(ops.pack_conv_weight(w)), t(b), t(sc), t(sh), t(r), 2 if res == 2 else 1, stride, pad, relu,
                     out_scale=t(osc), out_shift=t(osh))
    got = got.cpu().numpy()
    assert got.shape == want.shape
    scale = np.abs(want).max()
    err = np.abs(got - want).max()
    assert err <= (4e-5 if out_bn else 2e-5) * scale, "max err %g (scale %g)" % (err, scale)


# N,H,W,Cin,Cout,k,stride,pad, bias, residual, relu   -- ring-eligible shapes (no prologue, Cin % 32 == 0) whose K-step count is a
# multiple of 3: with split-K forced to 3 they run the in-workgroup form (conv_ring_kernel.h KG = 3: three 4-wave groups, LDS
# reduction, full epilogue by group 0)
KG_CASES = [
    (1, 9, 16, 96, 64, 1, 1, 0, False, 0, False),       # 1x1, 3 K-steps (one per group), a single ragged M tile
    (1, 17, 23, 192, 96, 1, 1, 0, True, 1, True),       # 1x1, 6 steps, ragged M and a ragged Cout tile, bias + residual + ReLU
    (1, 30, 30, 64, 64, 3, 1, 1, False, 0, False),      # 3x3 SAME (zero-page taps), 18 steps
    (2, 36, 64, 64, 128, 3, 1, 1, True, 1, True),       # 3x3, 72 x 2 tiles
    (1, 37, 63, 128, 128, 3, 2, 1, True, 0, True),      # 3x3 stride 2, odd sizes, 36 steps
    (1, 60, 60, 256, 256, 3, 1, 1, False, 1, False),    # the block-3 shape of a 720p frame: 57 x 4 tiles, 72 steps
    (1, 150, 160, 96, 64, 1, 1, 0, True, 0, False),     # 375 tiles > 256 CUs: several tiles per workgroup (ring across tile boundaries)
]


@pytest.mark.parametrize("out_bn", [False, True])
@pytest.mark.parametrize("N,H,W,Cin,Cout,k,stride,pad,bias,res,relu", KG_CASES)
def test_split_k_inside_the_workgroup(cuda, N, H, W, Cin, Cout, k, stride, pad, bias, res, relu, out_bn):
    from stabnet_amd import _lib, ops
    rng = np.random.default_rng(Cin * 5 + Cout + k + H)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32) if bias else None
    want = O.conv2d(x, w, stride, ((pad, pad), (pad, pad)), b)
    r = None
    if res:
        r = rng.standard_normal(want.shape).astype(np.float32)
        want = want + r
    osc = osh = None
    if out_bn:
        osc = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
        osh = (rng.standard_normal(Cout) * 0.3).astype(np.float32)
        want = (want * osc + osh).astype(np.float32)
    if relu:
        want = np.maximum(want, 0)
    t = lambda v: None if v is None else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(cuda)
    args = (t(x), t(ops.pack_conv_weight(w)), t(b), None, None, t(r), 1, stride, pad, relu)
    L = _lib.lib()
    try:
        L.stabnet_conv_tuning_override(2, 3)             # the 64 x 64 tile, three K slices
        got3 = ops.conv2d(*args, out_scale=t(osc), out_shift=t(osh)).cpu().numpy()
        got3b = ops.conv2d(*args, out_scale=t(osc), out_shift=t(osh)).cpu().numpy()
        L.stabnet_conv_tuning_override(2, 1)
        got1 = ops.conv2d(*args, out_scale=t(osc), out_shift=t(osh)).cpu().numpy()
    finally:
        L.stabnet_conv_tuning_override(-1, -1)
    scale = np.abs(want).max()
    for got in (got3, got1):
        assert got.shape == want.shape and np.isfinite(got).all()
        assert np.abs(got - want).max() <= (4e-5 if out_bn else 2e-5) * scale
    assert np.array_equal(got3, got3b)                   # groups are added in group order: the same bits every time
    assert np.abs(got3 - got1).max() <= 1e-5 * scale     # three slices vs one: float32 summation order only


# two K slices inside the workgroup WITH the fragment prologue (conv_ring_f32_kernel<0, 0, 2, 1>): the 1x1 layers that carry a
# BN + ReLU prologue and split K in two -- block-3 conv1 of a 720p frame (M = 3600, N = 256, K = 1024) and smaller relatives
KG2_CASES = [
    (1, 9, 16, 64, 64, False, 0, False),          # 2 K-steps: one per group, a single ragged tile
    (1, 17, 23, 192, 96, True, 1, True),          # 6 steps, ragged M and Cout tiles, bias + residual + ReLU
    (1, 45, 80, 1024, 256, False, 0, False),      # the 720p block-3 conv1 shape: 57 x 4 tiles, 16 steps per group
    (1, 150, 160, 128, 64, True, 0, True),        # 375 tiles > 256 CUs: the rings run across tile boundaries
    (2, 20, 24, 160, 64, False, 0, False),        # 5 steps: the slices are unequal -> stays on the slab path (still correct)
]


@pytest.mark.parametrize("out_bn", [False, True])
@pytest.mark.parametrize("N,H,W,Cin,Cout,bias,res,relu", KG2_CASES)
def test_two_k_groups_with_prologue(cuda, N, H, W, Cin, Cout, bias, res, relu, out_bn):
    from stabnet_amd import _lib, ops
    rng = np.random.default_rng(Cin * 3 + Cout + H)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((1, 1, Cin, Cout)) * np.sqrt(2.0 / Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32) if bias else None
    sc = rng.uniform(0.5, 1.5, Cin).astype(np.float32)
    sh = (rng.standard_normal(Cin) * 0.3).astype(np.float32)
    a = np.maximum(x * sc + sh, 0).astype(np.float32)
    want = O.conv2d(a, w, 1, ((0, 0), (0, 0)), b)
    r = None
    if res:
        r = rng.standard_normal(want.shape).astype(np.float32)
        want = want + r
    osc = osh = None
    if out_bn:
        osc = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
        osh = (rng.standard_normal(Cout) * 0.3).astype(np.float32)
        want = (want * osc + osh).astype(np.float32)
    if relu:
        want = np.maximum(want, 0)
    t = lambda v: None if v is None else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(cuda)
    args = (t(x), t(ops.pack_conv_weight(w)), t(b), t(sc), t(sh), t(r), 1, 1, 0, relu)
    L = _lib.lib()
    try:
        L.stabnet_conv_tuning_override(2, 2)             # the 64 x 64 tile, two K slices
        got2 = ops.conv2d(*args, out_scale=t(osc), out_shift=t(osh)).cpu().numpy()
        got2b = ops.conv2d(*args, out_scale=t(osc), out_shift=t(osh)).cpu().numpy()
        L.stabnet_conv_tuning_override(2, 1)
        got1 = ops.conv2d(*args, out_scale=t(osc), out_shift=t(osh)).cpu().numpy()
    finally:
        L.stabnet_conv_tuning_override(-1, -1)
    scale = np.abs(want).max()
    for got in (got2, got1):
        assert got.shape == want.shape and np.isfinite(got).all()
        assert np.abs(got - want).max() <= (4e-5 if out_bn else 2e-5) * scale
    assert np.array_equal(got2, got2b)
    assert np.abs(got2 - got1).max() <= 1e-5 * scale


@pytest.mark.parametrize("split", [1, 2])
def test_prologue_vectors_in_any_address_order(cuda, split):
    """The ABI takes in_scale and in_shift as two independent pointers.  The LDS-DMA kernel's fragment-prologue form reaches the
    shifts through a 32-bit unsigned offset from the scales, which only exists when the shifts lie 0..4 GiB above them: any other
    placement must take the register-staged kernel and give the same numbers (it used to wrap -> out-of-bounds read)."""
    from stabnet_amd import _lib, ops
    N, H, W, Cin, Cout = 1, 45, 80, 256, 64
    rng = np.random.default_rng(77)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((1, 1, Cin, Cout)) * np.sqrt(2.0 / Cin)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, Cin).astype(np.float32)
    sh = (rng.standard_normal(Cin) * 0.3).astype(np.float32)
    want = O.conv2d(np.maximum(x * sc + sh, 0).astype(np.float32), w, 1, ((0, 0), (0, 0)), None)
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(cuda)
    buf = torch.empty(4 * Cin, dtype=torch.float32, device=cuda)
    lo, hi = buf[:Cin], buf[3 * Cin:]
    L = _lib.lib()
    got = {}
    try:
        L.stabnet_conv_tuning_override(2, split)
        for name, (s_t, h_t) in {"shift_above": (lo, hi), "shift_below": (hi, lo)}.items():
            s_t.copy_(t(sc)); h_t.copy_(t(sh))
            assert (h_t.data_ptr() > s_t.data_ptr()) == (name == "shift_above")
            got[name] = ops.conv2d(t(x), t(ops.pack_conv_weight(w)), None, s_t, h_t, None, 1, 1, 0, False).cpu().numpy()
    finally:
        L.stabnet_conv_tuning_override(-1, -1)
    scale = np.abs(want).max()
    for g in got.values():
        assert np.isfinite(g).all() and np.abs(g - want).max() <= 2e-5 * scale
    assert np.abs(got["shift_above"] - got["shift_below"]).max() <= 1e-5 * scale    # two kernels, float32 summation order only


# conv2 (3x3) -> bn2 + ReLU -> conv3 (1x1) as ONE launch (conv_b2b_kernel.h; the block-1 / block-2 units of an inference frame).
# N,H,W,C,Cout,stride, bias, residual (0 none, 1 same size, 2 strided identity shortcut), relu, x_ch0 / extra channels of the input buffer
B2B_CASES = [
    (1, 9, 16, 64, 256, 1, True, 1, False, 0, 0),         # 3 ragged-free tiles... 144 px: 2 full tiles + a ragged one (64-row tiles)
    (1, 17, 23, 64, 256, 1, True, 1, True, 0, 0),         # ragged last tile, image borders everywhere
    (2, 36, 64, 64, 256, 1, True, 1, False, 0, 0),        # 72 tiles, two images (taps must not cross the image boundary)
    (1, 37, 63, 64, 256, 2, True, 2, False, 0, 0),        # stride 2, odd sizes, strided identity shortcut (block-1 unit 3)
    (1, 36, 64, 64, 256, 1, True, 1, False, 256, 0),      # input = channels 256..319 of a 320-channel buffer (merged shortcut|conv1)
    (1, 36, 64, 64, 256, 1, True, 1, False, 256, 1),      # ... and the residual = channels 0..255 of the SAME buffer (projection unit)
    (1, 36, 64, 64, 128, 1, False, 0, True, 0, 0),        # Cout = 2 chunks, no bias, no residual
    (1, 180, 320, 64, 256, 1, True, 1, False, 0, 0),      # the 720p block-1 shape: 900 tiles > 512 resident workgroups
    (1, 9, 16, 128, 512, 1, True, 1, False, 0, 0),        # d_b = 128: the 8-wave form
    (1, 23, 17, 128, 512, 1, True, 1, True, 0, 0),        # ragged
    (1, 37, 63, 128, 512, 2, True, 2, False, 0, 0),       # stride 2 (block-2 unit 4)
    (1, 36, 64, 128, 512, 1, True, 1, False, 512, 1),     # merged buffer, 640 channels
    (1, 90, 160, 128, 512, 1, True, 1, False, 0, 0),      # the 720p block-2 shape: 225 tiles, one per CU
    (2, 90, 160, 128, 512, 1, True, 1, False, 0, 0),      # 450 tiles: two per workgroup, the ring running across the tile boundary
]


@pytest.mark.parametrize("out_bn", [False, True])
@pytest.mark.parametrize("N,H,W,C,Cout,stride,bias,res,relu,x_ch0,res_in_x", B2B_CASES)
def test_conv3x3_conv1x1_one_launch(cuda, N, H, W, C, Cout, stride, bias, res, relu, x_ch0, res_in_x, out_bn):
    """vs the oracle's two convolutions, and vs the library's own two launches (same K order, no split: bit-identical)."""
    from stabnet_amd import _lib, ops
    rng = np.random.default_rng(C * 11 + Cout + H + stride)
    Cx = x_ch0 + C
    xfull = rng.standard_normal((N, H, W, Cx)).astype(np.float32)
    x = np.ascontiguousarray(xfull[..., x_ch0:])
    w2 = (rng.standard_normal((3, 3, C, C)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
    w3 = (rng.standard_normal((1, 1, C, Cout)) * np.sqrt(2.0 / C)).astype(np.float32)
    b3 = rng.standard_normal(Cout).astype(np.float32) if bias else None
    msc = rng.uniform(0.5, 1.5, C).astype(np.float32)
    msh = (rng.standard_normal(C) * 0.3).astype(np.float32)
    mid = O.conv2d(x, w2, stride, ((1, 1), (1, 1)), None)
    mid = np.maximum(mid * msc + msh, 0).astype(np.float32)
    want = O.conv2d(mid, w3, 1, ((0, 0), (0, 0)), b3)
    Ho, Wo = want.shape[1:3]
    r = None
    if res_in_x:
        assert res == 1 and stride == 1 and x_ch0 >= Cout
        r = xfull                                                        # channels 0 .. Cout of the input buffer
        want = want + xfull[..., :Cout]
    elif res == 1:
        r = rng.standard_normal((N, Ho, Wo, Cout)).astype(np.float32)
        want = want + r
    elif res == 2:
        r = rng.standard_normal((N, 2 * Ho - 1, 2 * Wo, Cout)).astype(np.float32)
        want = want + r[:, ::2, ::2, :]
    osc = osh = None
    if out_bn:
        osc = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
        osh = (rng.standard_normal(Cout) * 0.3).astype(np.float32)
        want = (want * osc + osh).astype(np.float32)
    if relu:
        want = np.maximum(want, 0)
    t = lambda v: None if v is None else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(cuda)
    xg = t(xfull)
    rg = xg if res_in_x else t(r)
    W2, W3 = t(ops.pack_conv_weight(w2)), t(ops.pack_conv_weight(w3))
    got = ops.conv3x3_conv1x1(xg, W2, t(msc), t(msh), W3, t(b3), rg, 2 if res == 2 else 1, stride, relu, t(osc), t(osh), x_ch0=x_ch0)
    got_b = ops.conv3x3_conv1x1(xg, W2, t(msc), t(msh), W3, t(b3), rg, 2 if res == 2 else 1, stride, relu, t(osc), t(osh), x_ch0=x_ch0)
    got, got_b = got.cpu().numpy(), got_b.cpu().numpy()
    assert got.shape == want.shape and np.isfinite(got).all()
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= (6e-5 if out_bn else 3e-5) * scale, np.abs(got - want).max() / scale
    assert np.array_equal(got, got_b)
    # the two-launch form of the same tail through the public conv operator, K unsplit: the same products in the same order
    L = _lib.lib()
    try:
        L.stabnet_conv_tuning_override(2, 1)
        mid_g = ops.conv2d(t(x), W2, None, None, None, None, 1, stride, 1, True, out_scale=t(msc), out_shift=t(msh))
        rr = t(np.ascontiguousarray(xfull[..., :Cout])) if res_in_x else t(r)
        two = ops.conv2d(mid_g, W3, t(b3), None, None, rr, 2 if res == 2 else 1, 1, 0, relu, out_scale=t(osc), out_shift=t(osh))
    finally:
        L.stabnet_conv_tuning_override(-1, -1)
    assert np.array_equal(got, two.cpu().numpy())
