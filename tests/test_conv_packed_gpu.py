"""GPU parity of the packed split convolution kernels (conv_ring_f32_kernel<MODE, 4, KG, PRO>, through stabnet_conv2d_fwd_packed):
float32 operands as exact sums of three bf16 terms, six bf16 x bf16 partial products per f32 product, f32 accumulation on
v_mfma_f32_32x32x16_bf16.  Same bar as the exact-f32-MFMA kernels (2e-5 of the output scale against the oracle's im2col + sgemm
convolution, different summation order) and a tighter one against those kernels themselves: this is not a reduced-precision mode."""
import numpy as np
import pytest
import torch

from oracle import stabnet_oracle as O

pytestmark = pytest.mark.gpu

# N,H,W,Cin,Cout,k,stride,pad, prologue, bias, residual(0 none,1 same,2 strided), relu, splitk (0 = planned)
CASES = [
    (1, 17, 23, 32, 64, 1, 1, 0, False, False, 0, False, 0),    # 1 K-step, ragged M
    (1, 17, 23, 96, 96, 1, 1, 0, False, True, 0, True, 0),      # 3 K-steps, ragged Cout tile
    (1, 17, 23, 128, 64, 1, 1, 0, False, False, 1, False, 0),   # 4 K-steps + residual
    (1, 17, 23, 160, 64, 1, 1, 0, False, False, 0, False, 0),   # 5 K-steps
    (1, 17, 23, 256, 64, 1, 1, 0, False, False, 0, False, 0),   # 8 K-steps: the mid-tile fast steps
    (2, 36, 64, 64, 64, 3, 1, 1, False, False, 0, False, 0),    # 3x3 SAME, zero page taps
    (1, 37, 63, 128, 128, 3, 2, 1, False, True, 0, True, 0),    # 3x3 stride 2, odd sizes
    (1, 36, 64, 64, 64, 3, 1, 2, False, False, 0, False, 0),    # pad 2
    (1, 9, 16, 512, 512, 3, 1, 1, False, False, 0, True, 0),    # planned split-K: slabs + reduce
    (1, 18, 32, 128, 512, 1, 2, 0, False, True, 2, False, 0),   # strided 1x1 + strided residual read
    (1, 180, 320, 64, 64, 3, 1, 1, False, True, 1, True, 0),    # 900 tiles > 512 resident workgroups: the ring across tile boundaries
    (1, 180, 320, 32, 256, 1, 1, 0, False, False, 1, False, 0), # 3600 one-step tiles
    (1, 60, 60, 256, 256, 3, 1, 1, False, True, 1, False, 3),   # the block-3 conv2 shape of a 720p frame, three slabs + reduce
    (1, 60, 60, 256, 256, 3, 1, 1, False, True, 1, True, 2),    # ... two K groups inside the workgroup
    (1, 30, 30, 64, 64, 3, 1, 1, False, False, 0, False, 2),    # 3x3, 18 steps, two groups, single ragged round
    (1, 17, 23, 192, 96, 1, 1, 0, False, True, 1, True, 2),     # 1x1, 6 steps, two groups, ragged M and Cout
    (1, 150, 160, 128, 64, 1, 1, 0, False, False, 0, False, 2), # 375 tiles > 256 CUs: several tiles per 8-wave workgroup
    # BN + ReLU prologue on the A fragments (the inference conv1 layers)
    (1, 36, 64, 64, 64, 1, 1, 0, True, False, 0, False, 0),
    (1, 36, 64, 256, 64, 1, 1, 0, True, True, 1, True, 0),
    (1, 90, 160, 512, 128, 1, 1, 0, True, False, 0, True, 0),   # block-2 conv1 of a 720p frame
    (1, 60, 60, 1024, 256, 1, 1, 0, True, False, 0, True, 2),   # block-3 conv1: prologue + two K groups inside the workgroup
    (1, 23, 40, 2048, 512, 1, 1, 0, True, False, 0, True, 4),   # block-4 conv1: prologue + slabs
    (1, 17, 23, 128, 96, 1, 1, 0, True, True, 0, False, 2),     # prologue, two groups, ragged tiles
    # geometries the packed kernels do not take: the same call runs the exact-f32 kernels on w_ohwi
    (1, 20, 24, 16, 64, 7, 2, 3, False, True, 0, False, 0),     # Cin = 16 (no weight image exists)
    (1, 37, 63, 128, 128, 3, 2, 1, True, False, 0, False, 0),   # 3x3 with a prologue (register-staged kernel)
]


@pytest.mark.parametrize("out_bn", [False, True])
@pytest.mark.parametrize("N,H,W,Cin,Cout,k,stride,pad,prologue,bias,res,relu,splitk", CASES)
def test_conv2d_packed_matches_oracle_and_f32_kernels(cuda, N, H, W, Cin, Cout, k, stride, pad, prologue, bias, res, relu, splitk, out_bn):
    from stabnet_amd import ops
    rng = np.random.default_rng(Cin * 7 + Cout + k + splitk)
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
    # scales and shifts as two slices of one buffer (the prologue kernels address the shifts relative to the scales)
    ss = t(np.concatenate([sc, sh])) if prologue else None
    tsc, tsh = (ss[:Cin], ss[Cin:]) if prologue else (None, None)
    args = (t(x), t(ops.pack_conv_weight(w)), t(b), tsc, tsh, t(r), 2 if res == 2 else 1, stride, pad, relu)
    got = ops.conv2d_packed(*args, out_scale=t(osc), out_shift=t(osh), splitk=splitk).cpu().numpy()
    f32 = ops.conv2d(*args, out_scale=t(osc), out_shift=t(osh)).cpu().numpy()
    assert got.shape == want.shape
    scale = np.abs(want).max()
    err = np.abs(got - want).max()
    assert err <= (4e-5 if out_bn else 2e-5) * scale, "max err vs oracle %g (scale %g)" % (err, scale)
    d = np.abs(got - f32).max()
    assert d <= 4e-6 * scale, "max difference to the exact-f32-MFMA kernels %g (scale %g)" % (d, scale)


def test_weight_image_is_an_exact_three_term_split(cuda):
    """Every weight equals h + m + l of its image entry, bit for bit (float32 sum of the three bf16 terms, small terms first)."""
    from stabnet_amd import _lib, ops
    from stabnet_amd._tensor import ptr, stream_ptr
    rng = np.random.default_rng(5)
    Cout, K = 96, 64
    w = (rng.standard_normal((Cout, 1, 1, K)) * np.exp(rng.uniform(-8, 8, (Cout, 1, 1, K)))).astype(np.float32)
    wt = torch.from_numpy(w).to(cuda)
    n = int(_lib.lib().stabnet_conv_weight_image_floats(Cout, 1, 1, K))
    assert n == 2 * 2 * 3072
    img = torch.zeros(n, dtype=torch.float32, device=cuda)
    _lib.call("stabnet_conv_weight_split_image", ptr(wt), Cout, 1, 1, K, ptr(img), stream_ptr(cuda), device=cuda)
    raw = img.cpu().numpy().view(np.uint16).reshape(2, 2, 2, 3, 2, 64, 8)          # [N tile][K step][wn][plane][j][lane][e]
    f = (raw.astype(np.uint32) << 16).view(np.float32)
    for n_ in range(Cout):
        nt, wn, ln = n_ // 64, (n_ % 64) // 32, n_ % 32
        for k in range(K):
            ks, kk = k // 32, k % 32
            j, rem = kk // 16, kk % 16
            e_hi, rem2 = rem // 8, rem % 8
            g, e_lo = rem2 // 4, rem2 % 4
            h, m, l = (f[nt, ks, wn, p, j, ln + 32 * g, 4 * e_hi + e_lo] for p in range(3))
            assert np.float32(np.float32(l + m) + h) == w[n_, 0, 0, k], (n_, k)
    # rows of the ragged second tile beyond Cout are zero
    assert not raw[1, :, 1].any()


@pytest.mark.parametrize("k,Cin,Cout", [(1, 512, 128), (3, 128, 64), (1, 2048, 64)])
def test_packed_error_against_float64_is_the_f32_mfma_error(cuda, k, Cin, Cout):
    """Not a reduced-precision mode: against a FLOAT64 convolution of the same float32 inputs, the packed split kernel's error is
    the exact-f32-MFMA kernel's error (both are float32 accumulations of exact -- resp. 2^-23-accurate -- products; the bf16-operand
    mode on the same data is three orders of magnitude further away)."""
    from stabnet_amd import ops
    rng = np.random.default_rng(k * 1000 + Cin)
    N, H, W = 1, 24, 32
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float32)
    pad = k // 2
    xp = np.pad(x.astype(np.float64), ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    cols = np.concatenate([xp[:, i:i + H, j:j + W, :] for i in range(k) for j in range(k)], axis=-1)      # [N,H,W,k*k*Cin] (kh, kw, c)
    want = cols.reshape(-1, k * k * Cin) @ w.astype(np.float64).reshape(k * k * Cin, Cout)
    want = want.reshape(N, H, W, Cout)
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(cuda)
    wt = t(ops.pack_conv_weight(w))
    got_p = ops.conv2d_packed(t(x), wt, pad=pad).cpu().numpy().astype(np.float64)
    got_f = ops.conv2d(t(x), wt, pad=pad).cpu().numpy().astype(np.float64)
    scale = np.abs(want).max()
    e_p, e_f = np.abs(got_p - want).max() / scale, np.abs(got_f - want).max() / scale
    r_p, r_f = np.sqrt(np.mean((got_p - want) ** 2)) / scale, np.sqrt(np.mean((got_f - want) ** 2)) / scale
    print("k=%d K=%d: max / rms error vs float64, relative to the output scale: packed split %.2e / %.2e, f32 MFMA %.2e / %.2e" % (
        k, k * k * Cin, e_p, r_p, e_f, r_f))
    assert e_p <= 2.0 * e_f + 1e-7 and r_p <= 1.5 * r_f + 2e-8
    assert e_p < 3e-6
