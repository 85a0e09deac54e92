"""GPU parity: HIP warp path (through the C ABI) vs the NumPy oracle -- bit-exact (integer/indices AND float32
values: the kernels follow the reference's op order with one rounding per op)."""
import numpy as np
import pytest
import torch

from oracle import stabnet_oracle as O

pytestmark = pytest.mark.gpu


def _cfg(H, W, gh=4, gw=4):
    from stabnet_amd.config import Config
    return Config(height=H, width=W, grid_h=gh, grid_w=gw), O.Config(height=H, width=W, grid_h=gh, grid_w=gw)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(got, want, what):
    got = np.asarray(got, np.float32).reshape(want.shape)
    neq = _bits(got) != _bits(want)
    # +0.0 / -0.0 are the same value
    neq &= ~((got == 0) & (want == 0))
    assert not neq.any(), "%s: %d of %d elements differ, max abs diff %g" % (
        what, int(neq.sum()), neq.size, float(np.abs(got - want)[neq].max()))


CASES = [
    # N, H, W, C, gh, gw, theta_std
    (2, 288, 512, 1, 4, 4, 0.05),
    (2, 32, 64, 1, 4, 4, 0.05),
    (2, 45, 77, 1, 4, 4, 0.08),     # remainder row/col, W % 4 != 0
    (1, 256, 256, 1, 4, 4, 0.0),    # identity mesh KAT
    (1, 96, 160, 3, 4, 4, 0.05),    # multi-channel
    (2, 64, 96, 1, 2, 3, 0.1),      # other grid
    (1, 72, 128, 1, 4, 4, 0.6),     # saturating clip +-1.25, folded cells
    (1, 720, 1280, 1, 4, 4, 0.05),  # BASELINE config 2 frame size
]


@pytest.mark.parametrize("N,H,W,C,gh,gw,std", CASES)
def test_warp_from_theta_bit_exact(cuda, N, H, W, C, gh, gw, std):
    from stabnet_amd import warp
    cfg, ocfg = _cfg(H, W, gh, gw)
    rng = np.random.default_rng(H * 1000 + W)
    theta = (rng.standard_normal((N, (gh + 1) * (gw + 1) * 2)) * std).astype(np.float32)
    U = (rng.random((N, H, W, C)) - 0.5).astype(np.float32)

    pts1, pts2 = O.get_4_pts(theta, ocfg)
    out, black, img, Hs, corners = O.transformer(U, pts2, ocfg, return_all=True)

    r = warp.warp_from_theta(torch.from_numpy(U).to(cuda), torch.from_numpy(theta).to(cuda), cfg)
    torch.cuda.synchronize()
    assert_bit_equal(r["pts2"].cpu().numpy(), pts2, "pts2")
    assert_bit_equal(r["Hs"].cpu().numpy(), Hs, "Hs")
    assert_bit_equal(r["x_map"].cpu().numpy(), img[..., 0], "x_map")
    assert_bit_equal(r["y_map"].cpu().numpy(), img[..., 1], "y_map")
    assert np.array_equal(r["black_pix"].cpu().numpy(), black), "black_pix"
    assert_bit_equal(r["output"].cpu().numpy(), out, "output")

    # reference-signature ops: get_4_pts + transformer
    p1, p2 = warp.get_4_pts(torch.from_numpy(theta).to(cuda), N, cfg)
    assert_bit_equal(p1.cpu().numpy(), pts1, "pts1")
    o2, b2, im2 = warp.transformer(torch.from_numpy(U).to(cuda), p2, cfg=cfg)
    assert_bit_equal(o2.cpu().numpy(), out, "transformer.output")
    assert_bit_equal(im2.cpu().numpy(), img, "transformer.img")
    assert np.array_equal(b2.cpu().numpy(), black)


def test_identity_mesh_known_answers(cuda):
    """SURVEY 8c KAT (1),(2),(8): theta=0 -> Hs = I +- ridge; x_map ~ linspace; last column samples ~0."""
    from stabnet_amd import warp
    H, W = 64, 128
    cfg, _ = _cfg(H, W)
    U = torch.ones((1, H, W, 1), device=cuda)
    r = warp.warp_from_theta(U, torch.zeros((1, 50), device=cuda), cfg)
    Hs = r["Hs"].cpu().numpy().reshape(16, 9)
    assert np.abs(Hs - np.eye(3).reshape(9)).max() < 1.2e-3
    xm = r["x_map"].cpu().numpy()[0, :, :, 0]
    assert np.abs(xm - np.linspace(-1, 1, W)[None, :]).max() < 2e-3
    out = r["output"].cpu().numpy()[0, :, :, 0]
    assert np.abs(out[1:-1, 1:-2] - 1.0).max() < 1e-4          # interior of a constant image stays constant
    # the ridge moves the outermost ring of pixels just past +-1 (strict test, :284): only that ring is black
    assert r["black_pix"][0, 1:-1, 1:-1].sum().item() == 0


@pytest.mark.parametrize("N,H,W,C", [(2, 288, 512, 1), (1, 45, 77, 2), (2, 64, 64, 1)])
def test_interpolate_bit_exact(cuda, N, H, W, C):
    from stabnet_amd import warp
    rng = np.random.default_rng(7)
    im = (rng.random((N, H, W, C)) - 0.5).astype(np.float32)
    gx, gy = np.meshgrid(np.linspace(-1, 1, W), np.linspace(-1, 1, H))
    x = (gx[None, :, :, None] + rng.normal(0, 0.05, (N, H, W, 1))).astype(np.float32)
    y = (gy[None, :, :, None] + rng.normal(0, 0.05, (N, H, W, 1))).astype(np.float32)
    x[0, 0, 0, 0] = 5.0       # far out of frame
    y[0, 1, 1, 0] = -7.0
    x[0, 2, 2, 0] = 1.0       # exactly on the border
    want = O.interpolate(im, x, y)
    got = warp.interpolate(torch.from_numpy(im).to(cuda), torch.from_numpy(x).to(cuda), torch.from_numpy(y).to(cuda),
                           (H, W))
    assert_bit_equal(got.cpu().numpy(), want, "interpolate")


def test_no_cpu_fallback(cuda):
    from stabnet_amd import warp
    from stabnet_amd._lib import StabnetError
    with pytest.raises(StabnetError):
        warp.interpolate(torch.zeros(1, 4, 4, 1), torch.zeros(1, 4, 4, 1), torch.zeros(1, 4, 4, 1), (4, 4))


def test_bad_args_report_error(cuda):
    from stabnet_amd import _lib
    rc = _lib.lib().stabnet_interp_fwd(0, 0, 0, 1, 4, 4, 1, 0, 0)
    assert rc == -1 and b"null" in _lib.lib().stabnet_last_error()
