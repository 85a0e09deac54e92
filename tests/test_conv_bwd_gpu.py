"""GPU: convolution backward (wgrad MFMA kernel, dgrad through the forward kernel) vs torch CPU float64 autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fnn

pytestmark = pytest.mark.gpu

# N,H,W,Cin,Cout,k,stride,pad,prologue
CASES = [
    (2, 18, 24, 64, 64, 1, 1, 0, True),
    (2, 18, 24, 64, 64, 3, 1, 1, True),
    (1, 19, 23, 64, 128, 3, 2, 1, True),      # strided conv2, odd sizes -> dilated dgrad
    (2, 9, 16, 256, 64, 1, 1, 0, True),
    (1, 20, 28, 16, 64, 7, 2, 3, False),      # stem (wgrad only is used by the net; dgrad checked too)
    (3, 12, 12, 128, 512, 1, 1, 0, False),
    # the stride-1 "same" wgrad kernel (scalar running bases): W = 32 (no column remainder), W = 16 < 32 (two image rows per
    # step), W = 64, several images per split, no prologue, K and Cout of several tiles
    (2, 16, 32, 64, 64, 3, 1, 1, True),
    (2, 8, 16, 128, 128, 3, 1, 1, True),
    (1, 32, 64, 64, 128, 3, 1, 1, False),
    (8, 8, 8, 128, 192, 1, 1, 0, True),
    (6, 8, 16, 64, 64, 3, 1, 1, True),
    # stride-2 dgrad with whole output rows per tile (dx width % 64 == 0): the tap rows that only meet the dilation's zeros are skipped
    (2, 16, 128, 64, 64, 3, 2, 1, True),
    (1, 8, 64, 128, 128, 3, 2, 1, False),
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,stride,pad,pro", CASES)
def test_conv_backward(cuda, N, H, W, Cin, Cout, k, stride, pad, pro):
    from stabnet_amd import ops
    rng = np.random.default_rng(Cin + Cout + k)
    x = rng.standard_normal((N, H, W, Cin))
    w = rng.standard_normal((Cout, k, k, Cin)) * np.sqrt(2.0 / (k * k * Cin))
    sc = rng.uniform(0.5, 1.5, Cin) if pro else None
    sh = rng.standard_normal(Cin) * 0.3 if pro else None
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    a = torch.relu(xt * torch.tensor(sc) + torch.tensor(sh)) if pro else xt
    a.retain_grad()
    y = Fnn.conv2d(a.permute(0, 3, 1, 2), wt.permute(0, 3, 1, 2), stride=stride, padding=pad).permute(0, 2, 3, 1)
    g = rng.standard_normal(tuple(y.shape))
    (y * torch.tensor(g)).sum().backward()
    f = lambda v: None if v is None else torch.tensor(np.ascontiguousarray(v), dtype=torch.float32, device=cuda)
    dw = ops.conv2d_wgrad(f(x), f(g), (Cout, k, k, Cin), f(sc), f(sh), stride, pad)
    want = wt.grad.numpy()
    assert np.abs(dw.cpu().numpy() - want).max() <= 2e-5 * np.abs(want).max() * np.sqrt(N * H * W / 64 + 1)
    if Cout % 16 == 0:
        dx = ops.conv2d_dgrad(f(g), f(w), (N, H, W, Cin), stride, pad)
        want = a.grad.numpy()                                   # gradient wrt the activated input of the conv
        assert np.abs(dx.cpu().numpy() - want).max() <= 2e-5 * np.abs(want).max()
        res = rng.standard_normal((N, H, W, Cin))
        dx2 = ops.conv2d_dgrad(f(g), f(w), (N, H, W, Cin), stride, pad, residual=f(res))
        assert np.abs(dx2.cpu().numpy() - (want + res)).max() <= 2e-5 * np.abs(want + res).max()
    # accumulation semantics of wgrad (second tower adds)
    dw2 = ops.conv2d_wgrad(f(x), f(g), (Cout, k, k, Cin), f(sc), f(sh), stride, pad, dw=dw.clone())
    assert torch.allclose(dw2, 2 * dw, rtol=1e-4, atol=1e-5 * float(dw.abs().max()))


# the 13-channel stem's weight gradient on the tight zero-bordered operand (filter-row runs): several images (the border is per
# image), odd sizes, another channel count / filter / stride, more than one tile of output channels
ROWRUN_CASES = [
    (1, 20, 28, 13, 16, 64, 7, 2, 3),
    (3, 21, 27, 13, 16, 64, 7, 2, 3),
    (2, 32, 64, 13, 16, 64, 7, 2, 3),
    (2, 16, 24, 7, 8, 128, 3, 1, 1),
    (2, 17, 40, 5, 5, 32, 5, 2, 2),
]


@pytest.mark.parametrize("N,H,W,Cin,CinPad,Cout,k,stride,pad", ROWRUN_CASES)
def test_wgrad_rowrun(cuda, N, H, W, Cin, CinPad, Cout, k, stride, pad):
    from stabnet_amd import ops
    rng = np.random.default_rng(N + H + Cin)
    x = rng.standard_normal((N, H, W, Cin))
    xt = torch.tensor(x, dtype=torch.float64)
    wt = torch.zeros((Cout, k, k, Cin), dtype=torch.float64, requires_grad=True)
    y = Fnn.conv2d(xt.permute(0, 3, 1, 2), wt.permute(0, 3, 1, 2), stride=stride, padding=pad).permute(0, 2, 3, 1)
    g = rng.standard_normal(tuple(y.shape))
    (y * torch.tensor(g)).sum().backward()
    want = wt.grad.numpy()
    f = lambda v: torch.tensor(np.ascontiguousarray(v), dtype=torch.float32, device=cuda)
    seed = f(rng.standard_normal((Cout, k, k, CinPad)))          # accumulated into; pad channels stay what they were
    dw = ops.conv2d_wgrad_rowrun(f(x), f(g), (Cout, k, k, CinPad), stride, pad, dw=seed.clone())
    got = (dw - seed).cpu().numpy()
    assert np.abs(got[..., :Cin] - want).max() <= 2e-5 * np.abs(want).max() * np.sqrt(N * H * W / 64 + 1) + 1e-6
    assert np.array_equal(dw[..., Cin:].cpu().numpy(), seed[..., Cin:].cpu().numpy())
    # against the library's general kernel on the channel-padded input: same products, another summation order
    xp = np.zeros((N, H, W, 16), dtype=np.float32)
    xp[..., :Cin] = x
    ref = ops.conv2d_wgrad(f(xp), f(g), (Cout, k, k, 16), None, None, stride, pad).cpu().numpy()[..., :Cin]
    assert np.abs(got[..., :Cin] - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-6


# the unit-closing 1x1 layers: the bias gradient rides in the wgrad launch (column sums of dy per pixel split, reduced with the slabs)
@pytest.mark.parametrize("N,H,W,Cin,Cout,pro", [(2, 16, 32, 64, 256, True), (2, 9, 16, 128, 512, True), (3, 8, 8, 512, 2048, False),
                                                (2, 72, 128, 64, 256, True)])
def test_wgrad_with_bias(cuda, N, H, W, Cin, Cout, pro):
    from stabnet_amd import ops
    rng = np.random.default_rng(Cin + Cout + N)
    x = rng.standard_normal((N, H, W, Cin))
    g = rng.standard_normal((N, H, W, Cout))
    sc = rng.uniform(0.5, 1.5, Cin) if pro else None
    sh = rng.standard_normal(Cin) * 0.3 if pro else None
    f = lambda v: None if v is None else torch.tensor(np.ascontiguousarray(v), dtype=torch.float32, device=cuda)
    dw0 = f(rng.standard_normal((Cout, 1, 1, Cin)))
    db0 = f(rng.standard_normal(Cout))
    dw, db = ops.conv2d_wgrad_bias(f(x), f(g), (Cout, 1, 1, Cin), f(sc), f(sh), dw=dw0, db=db0)
    ref = ops.conv2d_wgrad(f(x), f(g), (Cout, 1, 1, Cin), f(sc), f(sh), 1, 0, dw=dw0.clone())
    assert torch.equal(dw, ref)                                 # the weight gradient is the plain launch's, bit for bit
    want = g.reshape(-1, Cout).sum(0)
    got = (db - db0).cpu().numpy()
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max() * np.sqrt(N * H * W / 64 + 1)


def test_wgrad_with_bias_refuses_other_geometries(cuda):
    from stabnet_amd import ops, _lib
    x = torch.zeros((1, 8, 8, 64), device=cuda)
    g = torch.zeros((1, 8, 8, 64), device=cuda)
    with pytest.raises(_lib.StabnetError):
        ops.conv2d_wgrad_bias(x, g, (64, 1, 1, 64))             # Cout % 256 != 0
