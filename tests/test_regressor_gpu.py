"""GPU parity of the whole regressor (resnet_v2_50 -> mean -> FC head, moving-average BN) against the oracle.
Float32 both sides, different summation orders: tolerance is stated per check."""
import numpy as np
import pytest
import torch

from oracle import stabnet_oracle as O

pytestmark = pytest.mark.gpu

TAPS = ["conv1", "pool1", "block1/unit_1", "block1/unit_3", "block2/unit_4", "block3/unit_6", "block4/unit_3",
        "global_pool"]


def _setup(N, H, W):
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    cfg = Config(height=H, width=W)
    ocfg = O.Config(height=H, width=W)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
    x, _ = synthetic.make_stack(cfg, N, H, W, seed=3)
    return cfg, ocfg, P, x


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (1, 90, 130), (1, 288, 512)])
def test_regressor_taps_and_theta(cuda, N, H, W):
    from stabnet_amd.regressor import Regressor
    cfg, ocfg, P, x = _setup(N, H, W)
    taps = {}
    theta_ref, _, _ = O.get_resnet(x, P, ocfg, taps=taps)
    reg = Regressor(P, N, H, W, cfg, keep_activations=True)
    theta = reg(torch.from_numpy(x).to(cuda))
    torch.cuda.synchronize()
    for name in TAPS:
        got = reg.activation(name).cpu().numpy()
        want = taps[name].reshape(got.shape)
        scale = np.abs(want).max()
        err = np.abs(got - want).max()
        assert err <= 5e-5 * scale, "%s: max err %g vs scale %g" % (name, err, scale)
    err = np.abs(theta.cpu().numpy() - theta_ref).max()
    assert err <= 2e-5, "theta max err %g (theta scale %g)" % (err, np.abs(theta_ref).max())

    # same result with activation-buffer reuse (the deploy configuration)
    reg2 = Regressor(P, N, H, W, cfg, keep_activations=False)
    theta2 = reg2(torch.from_numpy(x).to(cuda))
    # (the inference plan's stem sums its taps in a different order: 13-channel row runs instead of 16-channel pixels)
    assert (theta - theta2).abs().max().item() <= 2e-6


def test_full_frame_matches_oracle(cuda):
    """x_tensor -> (output_img, black_pix, Hs, x_map, y_map): the fetch list of deploy_bundle.py:286."""
    from stabnet_amd.regressor import Regressor
    from stabnet_amd import warp
    N, H, W = 1, 288, 512
    cfg, ocfg, P, x = _setup(N, H, W)
    ref = O.inference_stable_net(x, P, ocfg)
    reg = Regressor(P, N, H, W, cfg)
    xt = torch.from_numpy(x).to(cuda)
    theta = reg(xt)
    cur = xt[..., 2 * cfg.before_ch:2 * cfg.before_ch + 1].contiguous()
    r = warp.warp_from_theta(cur, theta, cfg)
    xm, ym = r["x_map"].cpu().numpy(), r["y_map"].cpu().numpy()
    # homography tolerance is stated on the maps: 1e-4 normalised = 0.026 px at W=512
    assert np.abs(xm - ref["x_map"]).max() < 1e-4 and np.abs(ym - ref["y_map"]).max() < 1e-4
    # black_pix may flip only where the map is within the tolerance of +-1
    flips = r["black_pix"].cpu().numpy() != ref["black_pix"]
    edge = (np.abs(np.abs(ref["x_map"][..., 0]) - 1) < 1e-4) | (np.abs(np.abs(ref["y_map"][..., 0]) - 1) < 1e-4)
    assert not (flips & ~edge).any()
    # warped pixels: bilinear sampling is Lipschitz in the sample position with constant <= 2*G per pixel of
    # displacement (G = largest neighbour difference of the source frame), so the map tolerance bounds the pixel error
    src = x[0, :, :, 2 * cfg.before_ch]
    G = max(np.abs(np.diff(src, axis=0)).max(), np.abs(np.diff(src, axis=1)).max())
    dpx = np.abs(xm - ref["x_map"]) * W / 2 + np.abs(ym - ref["y_map"]) * H / 2
    bound = 2 * G * dpx[..., 0] + 1e-5
    # ... except where the sample sits on the frame border: there the reference's clipped-corner weights make the
    # sampler discontinuous (in-frame value vs ~0), so a 1e-4 map difference may legitimately flip the pixel
    xp = (ref["x_map"][..., 0] + 1) * W / 2
    yp = (ref["y_map"][..., 0] + 1) * H / 2
    tol = 0.05
    border = (np.abs(xp) < tol) | (np.abs(xp - (W - 1)) < tol) | (np.abs(yp) < tol) | (np.abs(yp - (H - 1)) < tol)
    err = np.abs(r["output"].cpu().numpy() - ref["output"])[..., 0]
    assert (err <= bound)[~border].all(), "max excess %g" % float((err - bound)[~border].max())
    # given the oracle's theta, everything downstream is bit-exact
    r2 = warp.warp_from_theta(cur, torch.from_numpy(ref["theta"]).to(cuda), cfg)
    assert np.array_equal(r2["output"].cpu().numpy(), ref["output"])
    assert np.array_equal(r2["black_pix"].cpu().numpy(), ref["black_pix"])


def test_odd_size_with_poisoned_workspace(cuda):
    """45x77 (odd x odd: the stem's last 32-float row run of the last image reads past its taps into the slack row) with the
    whole workspace pre-filled with NaN: the over-read meets zero weights, and 0 * NaN would poison theta unless the slack
    row is zeroed by the stack assembly / border embedding (both entries are exercised)."""
    from stabnet_amd.deploy import StabNetStream
    from stabnet_amd.regressor import Regressor
    N, H, W = 1, 45, 77
    cfg, ocfg, P, x = _setup(N, H, W)
    theta_ref, _, _ = O.get_resnet(x, P, ocfg)
    reg = Regressor(P, N, H, W, cfg)
    reg.workspace.view(torch.float32).fill_(float("nan"))
    theta = reg(torch.from_numpy(x).to(cuda))
    torch.cuda.synchronize()
    assert torch.isfinite(theta).all()
    assert np.abs(theta.cpu().numpy() - theta_ref).max() <= 2e-5
    # the deploy entry (stack assembled from the ring) on the same odd size
    s = StabNetStream(P, H, W, cfg, streams=1, device=cuda)
    s.reg.workspace.view(torch.float32).fill_(float("nan"))
    clip = torch.from_numpy(x[0, :, :, 12]).to(cuda)[None]
    s.start(clip)
    r = s.step(clip)
    torch.cuda.synchronize()
    assert torch.isfinite(r["theta"]).all() and torch.isfinite(r["output"]).all()


def test_wrong_pointer_is_an_error_not_a_fault(cuda):
    """The C entry points check that the buffers belong to the current device (a host pointer, or memory of another GPU
    with the wrong device current, must come back as an error code instead of faulting inside a kernel)."""
    from stabnet_amd import _lib
    from stabnet_amd.regressor import Regressor
    N, H, W = 1, 64, 96
    cfg, ocfg, P, x = _setup(N, H, W)
    reg = Regressor(P, N, H, W, cfg)
    host_params = reg.params.cpu()                              # pageable host memory: not a device pointer
    theta = torch.empty((N, cfg.n_theta), device=cuda)
    xt = torch.from_numpy(x).to(cuda)
    with pytest.raises(_lib.StabnetError, match="device"):
        _lib.call("stabnet_backbone_fwd_infer", reg.plan.handle, host_params.data_ptr(), reg.fold.data_ptr(), xt.data_ptr(),
                  theta.data_ptr(), reg.workspace.data_ptr(), reg.workspace.numel(), torch.cuda.current_stream().cuda_stream, 0)


def test_get_resnet_training_and_inference_branches(cuda):
    """get_resnet(x, reuse, is_training, n) (s_net_bundle_nobm.py:250-264): is_training=True is the batch-statistics branch of
    :301 (and updates the moving averages, slim UPDATE_OPS), False the moving-average branch of :302; both return
    (theta, id2_loss, id2_loss) with id2_loss = mean|theta| * id_mul (:263-264)."""
    from stabnet_amd.regressor import Regressor, get_resnet
    N, H, W = 2, 64, 96
    cfg, ocfg, P, x = _setup(N, H, W)
    xt = torch.from_numpy(x).to(cuda)
    reg = Regressor(P, N, H, W, cfg)
    th_i, id_i, id2_i = get_resnet(xt, None, False, N, regressor=reg)
    ref_i, _, rid_i = O.get_resnet(x, P, ocfg, training=False)
    assert np.abs(th_i.cpu().numpy() - ref_i).max() <= 2e-5
    assert float(id2_i) == pytest.approx(float(rid_i), rel=1e-4) and float(id_i) == float(id2_i)
    mov0 = reg.params[reg.plan.n_trainable:].clone()
    stats = {}
    th_t, _, id2_t = get_resnet(xt, None, True, N, regressor=reg)
    ref_t, _, rid_t = O.get_resnet(x, P, ocfg, training=True, stats_out=stats)
    assert np.abs(th_t.cpu().numpy() - ref_t).max() <= 5e-5
    assert float(id2_t) == pytest.approx(float(rid_t), rel=1e-3)
    assert np.abs(ref_t - ref_i).max() > 1e-3                       # the two branches do differ on this input
    # the moving averages moved towards the batch statistics with decay 0.997
    q = reg.plan.unpack(reg.params.cpu().numpy())
    name = "resnet_v2_50/block2/unit_1/bottleneck_v2/conv1/BatchNorm"
    want = P[name + "/moving_variance"] - (P[name + "/moving_variance"] - stats[name][1]) * (1 - cfg.bn_decay)
    assert np.abs(q[name + "/moving_variance"] - want).max() < 1e-5
    assert not torch.equal(mov0, reg.params[reg.plan.n_trainable:])


def test_mirror_repacking_kernels(cuda):
    """x_tensor[..., 12:13] and img = [x_map, y_map] are library kernels (no torch arithmetic on the path)."""
    from stabnet_amd import warp
    from stabnet_amd.config import Config
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 9, 11, 13, generator=g).to(cuda)
    for c in (0, 12):
        assert torch.equal(warp.slice_channel(x, c), x[..., c:c + 1])
    cfg = Config(height=32, width=64)
    theta = (torch.randn(2, 50, generator=g) * 0.05).to(cuda)
    U = torch.rand(2, 32, 64, 1, generator=g).to(cuda)
    pts1, pts2, Hs = warp.get_4_pts(theta, 2, cfg, with_Hs=True)
    out, black, img, Hs2 = warp.transformer(U, pts2, cfg=cfg, return_Hs=True)
    r = warp.warp_from_theta(U, theta, cfg)
    assert torch.equal(img[..., 0], r["x_map"][..., 0]) and torch.equal(img[..., 1], r["y_map"][..., 0]) and torch.equal(out, r["output"])
    ocfg = O.Config(height=32, width=64)
    p1, p2 = O.get_4_pts(theta.cpu().numpy(), ocfg)
    assert np.array_equal(pts1.cpu().numpy(), p1) and np.array_equal(pts2.cpu().numpy(), p2)
