"""GPU: backward of warp / sampler / losses (C ABI) against the torch float64 autograd oracle.
Tolerance 2e-3 of the gradient scale: float32 kernels with atomics vs float64."""
import numpy as np
import pytest
import torch

from oracle import stabnet_oracle as O
from oracle import torch_ref as T

pytestmark = pytest.mark.gpu


def rel(got, want):
    want = np.asarray(want, np.float64)
    return float(np.abs(np.asarray(got, np.float64).reshape(want.shape) - want).max() / (np.abs(want).max() + 1e-30))


@pytest.mark.parametrize("N,H,W,C,std", [(2, 32, 64, 1, 0.05), (2, 45, 77, 1, 0.08), (1, 288, 512, 1, 0.05), (1, 40, 48, 2, 0.05)])
def test_transformer_and_mesh_losses_backward(cuda, N, H, W, C, std):
    from stabnet_amd import train_ops, warp
    from stabnet_amd.config import Config
    cfg = Config(height=H, width=W, batch_size=N)
    ocfg = O.Config(height=H, width=W, batch_size=N)
    rng = np.random.default_rng(H + W)
    theta = (rng.standard_normal((N, 50)) * std).astype(np.float32)
    theta[0, 0] = -0.4                                     # saturated vertex -> zero gradient through the clip
    U = (rng.random((N, H, W, C)) - 0.5).astype(np.float32)
    g_out = rng.standard_normal((N, H, W, C)).astype(np.float32)
    g_x = rng.standard_normal((N, H, W)).astype(np.float32)
    g_y = rng.standard_normal((N, H, W)).astype(np.float32)
    w_id, w_dist, w_cons = 0.16, 1.0, 20.0

    th = T.t(theta, requires_grad=True)
    pts1, pts2 = T.get_4_pts(th, ocfg)
    # sampler corners as decided in float32 by the forward (floor has no gradient; see torch_ref._sample)
    _, np_pts2 = O.get_4_pts(theta, ocfg)
    corners = O.transformer(U, np_pts2, ocfg, return_all=True)[4]
    out, black, flow, Hs = T.transformer(T.t(U), pts2, ocfg, corners)
    id2 = th.abs().mean() * ocfg.id_mul
    L = (out * T.t(g_out)).sum() + (flow[..., 0] * T.t(g_x)).sum() + (flow[..., 1] * T.t(g_y)).sum() \
        + w_dist * T.get_distortion_loss(pts1, ocfg) + w_cons * T.get_consistency_loss(pts2, ocfg) + w_id * id2
    L.backward()
    want = th.grad.numpy()

    dev = lambda a: torch.from_numpy(a).to(cuda)
    r = warp.warp_from_theta(dev(U), dev(theta), cfg)
    d_pts2 = train_ops.transformer_bwd(r["pts2"], r["Hs"], dev(U), r["x_map"], r["y_map"], dev(g_out), dev(g_x), dev(g_y), cfg)
    losses, d_theta = train_ops.mesh_losses(dev(theta), d_pts2, cfg, w_id, w_dist, w_cons, 1.0, 120.0)
    assert rel(d_theta.cpu().numpy(), want) < 2e-3
    assert d_theta[0, 0].item() == pytest.approx(w_id * cfg.id_mul * -1.0 / (N * 50), rel=1e-5)   # clip kills the rest
    lv = losses.cpu().numpy()
    assert lv[0] == pytest.approx(float(id2), rel=1e-5)
    assert lv[1] == 0.0
    assert lv[2] == pytest.approx(float(T.get_distortion_loss(pts1, ocfg)), rel=1e-4)
    assert lv[3] == pytest.approx(float(T.get_consistency_loss(pts2, ocfg)), rel=1e-4)


# (near-identity flow, one tile) / several 16 x 128 tiles with halo traffic between them / a flow that sends every tap far from its
# tile (the kernel's direct-to-memory path) / two channels (the untiled kernel)
@pytest.mark.parametrize("N,H,W,C,flow", [(2, 45, 77, 1, 0.05), (2, 70, 300, 1, 0.05), (1, 70, 300, 1, None), (1, 33, 130, 2, 0.05)])
def test_interp_backward(cuda, N, H, W, C, flow):
    from stabnet_amd import train_ops
    rng = np.random.default_rng(4)
    im = rng.standard_normal((N, H, W, C)).astype(np.float32)
    gx, gy = np.meshgrid(np.linspace(-1, 1, W), np.linspace(-1, 1, H))
    if flow is None:
        x = rng.uniform(-1.2, 1.2, (N, H, W)).astype(np.float32)
        y = rng.uniform(-1.2, 1.2, (N, H, W)).astype(np.float32)
    else:
        x = (gx[None] + rng.normal(0, flow, (N, H, W))).astype(np.float32)
        y = (gy[None] + rng.normal(0, flow, (N, H, W))).astype(np.float32)
    g = rng.standard_normal((N, H, W, C)).astype(np.float32)
    imt = T.t(im, requires_grad=True)
    corners = O._interpolate(im, x.reshape(-1), y.reshape(-1))[1]
    (T.interpolate(imt, T.t(x), T.t(y), corners) * T.t(g)).sum().backward()
    dev = lambda a: torch.from_numpy(a).to(cuda)
    got = train_ops.interp_bwd(dev(x), dev(y), dev(g))
    assert rel(got.cpu().numpy(), imt.grad.numpy()) < 1e-4


def test_masked_mse_and_feature_loss(cuda):
    from stabnet_amd import train_ops
    N, H, W, Mx = 2, 32, 64, 50
    cfg = O.Config(height=H, width=W, batch_size=N, max_matches=Mx)
    rng = np.random.default_rng(9)
    a = rng.standard_normal((N, H, W, 1)).astype(np.float32)
    b = rng.standard_normal((N, H, W, 1)).astype(np.float32)
    black = (rng.random((N, H, W)) < 0.2).astype(np.float32)
    m2 = rng.random((N, H, W, 1)).astype(np.float32)
    dev = lambda v: torch.from_numpy(v).to(cuda)
    for mm in (None, m2):
        at, bt = T.t(a, True), T.t(b, True)
        keep = (1 - T.t(black)).reshape(N, H, W, 1) * (T.t(mm) if mm is not None else 1.0)
        err = (at - bt) * keep
        loss = ((err * err).sum(dim=(1, 2, 3)) / (keep.sum(dim=(1, 2, 3)) + 1e-8)).sum() / N
        (loss * 3.0).backward()
        sums = train_ops.masked_mse_sums(dev(a), dev(b), dev(black), dev(mm) if mm is not None else None)
        s = sums.cpu().numpy().astype(np.float64)
        assert float((s[:, 0] / (s[:, 1] + 1e-8)).sum() / N) == pytest.approx(float(loss), rel=1e-5)
        ga, gb = train_ops.masked_mse_grad(dev(a), dev(b), dev(black), dev(mm) if mm is not None else None, sums,
                                           3.0 / N, want_gb=True)
        assert rel(ga.cpu().numpy(), at.grad.numpy()) < 1e-4 and rel(gb.cpu().numpy(), bt.grad.numpy()) < 1e-4
    # feature loss
    matches = rng.uniform(-1.1, 1.1, (N, Mx, 4)).astype(np.float32)
    mask = (rng.random((N, Mx)) < 0.6).astype(np.float32)
    flow = rng.standard_normal((N, H, W, 2)).astype(np.float32)
    ft = T.t(flow, True)
    fl = T.feature_loss(T.t(matches), T.t(mask), ft, cfg)
    (fl * 2.5).backward()
    val, dxm, dym, warped, dscale = train_ops.feature_loss(dev(matches), dev(mask), dev(np.ascontiguousarray(flow[..., 0])),
                                                           dev(np.ascontiguousarray(flow[..., 1])), 2.5 / N, True, True)
    assert float(val.mean()) == pytest.approx(float(fl), rel=1e-5)
    # the map gradient comes as signed counts (exact small integers, whatever the order of the scatter) x a per-sample factor
    cnt = dxm.cpu().numpy()
    assert np.array_equal(cnt, np.rint(cnt)) and np.abs(cnt).max() >= 1
    gx = cnt * dscale.cpu().numpy()[:, None, None]
    gy = dym.cpu().numpy() * dscale.cpu().numpy()[:, None, None]
    assert rel(gx, ft.grad.numpy()[..., 0]) < 1e-5 and rel(gy, ft.grad.numpy()[..., 1]) < 1e-5
    want_warped, _ = O.warp_pts(matches[:, :, :2], flow, cfg)
    assert np.array_equal(warped.cpu().numpy(), want_warped)


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), 1e9])
def test_non_finite_gradients_propagate_through_the_fixed_point_sums(cuda, bad):
    """The order-independent 64-bit fixed-point accumulators cannot hold NaN / Inf / |v| >= 2^22: such a contribution poisons
    the sums and the gradients come out NaN (a diverging step must not hand Adam finite-looking garbage), while a clean call on
    the same buffers afterwards is finite again."""
    from stabnet_amd import train_ops, warp
    from stabnet_amd.config import Config
    N, H, W = 2, 32, 64
    cfg = Config(height=H, width=W, batch_size=N)
    rng = np.random.default_rng(0)
    dev = lambda a: torch.from_numpy(a).to(cuda)
    theta = dev((rng.standard_normal((N, 50)) * 0.05).astype(np.float32))
    U = dev((rng.random((N, H, W, 1)) - 0.5).astype(np.float32))
    r = warp.warp_from_theta(U, theta, cfg)
    g = rng.standard_normal((N, H, W, 1)).astype(np.float32)
    clean = train_ops.transformer_bwd(r["pts2"], r["Hs"], U, r["x_map"], r["y_map"], dev(g), None, None, cfg)
    assert torch.isfinite(clean).all()
    gb = g.copy()
    gb[1, 7, 9, 0] = bad
    d_pts2 = train_ops.transformer_bwd(r["pts2"], r["Hs"], U, r["x_map"], r["y_map"], dev(gb), None, None, cfg)
    assert torch.isnan(d_pts2).any(), "a %r in d_out left a finite d_pts2" % bad
    again = train_ops.transformer_bwd(r["pts2"], r["Hs"], U, r["x_map"], r["y_map"], dev(g), None, None, cfg)
    assert torch.equal(again, clean)
    # the flow sampler's scatter: d_im
    fx = dev(rng.uniform(-1, 1, (N, H, W)).astype(np.float32))
    fy = dev(rng.uniform(-1, 1, (N, H, W)).astype(np.float32))
    d_clean = train_ops.interp_bwd(fx, fy, dev(g))
    assert torch.isfinite(d_clean).all()
    d_im = train_ops.interp_bwd(fx, fy, dev(gb))
    assert torch.isnan(d_im).any(), "a %r in d_out left a finite d_im" % bad
    assert torch.equal(train_ops.interp_bwd(fx, fy, dev(g)), d_clean)
