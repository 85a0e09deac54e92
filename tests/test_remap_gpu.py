"""GPU: colour-frame remap with smoothed maps (deploy_bundle.py:136-146) vs the oracle's restatement of OpenCV's
resize/remap geometry -- smoothed pixel-coordinate maps bit-exact, uint8 output exact."""
import numpy as np
import pytest
import torch

from oracle import stabnet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shift", [0.0, 0.45, -0.45])      # +-0.45: a fifth of the frame maps outside on the right / bottom or left / top
@pytest.mark.parametrize("H,W", [(288, 512), (90, 130), (720, 1280), (64, 96)])   # W % 4 == 0: the four-pixels-per-thread kernel; 130: the scalar one
def test_warp_rev_bundle2(cuda, H, W, shift):
    from stabnet_amd import warp
    from stabnet_amd.config import Config
    cfg, ocfg = Config(height=H, width=W), O.Config(height=H, width=W)
    rng = np.random.default_rng(H)
    theta = (rng.standard_normal((1, 50)) * 0.06).astype(np.float32)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    _, pts2 = O.get_4_pts(theta, ocfg)
    x_map, y_map, _ = O.maps_from_Hs(O.get_Hs(pts2, ocfg), H, W, ocfg)
    x_map, y_map = (x_map + np.float32(shift)).astype(np.float32), (y_map + np.float32(shift)).astype(np.float32)
    want, xs, ys = O.warpRevBundle2(img, x_map[0], y_map[0])
    got, px, py = warp.warpRevBundle2(torch.from_numpy(img).to(cuda), torch.from_numpy(x_map).to(cuda),
                                      torch.from_numpy(y_map).to(cuda), return_maps=True)
    assert np.array_equal(px.cpu().numpy()[0], xs) and np.array_equal(py.cpu().numpy()[0], ys)
    assert np.array_equal(got.cpu().numpy(), want)
    if shift:
        assert (want == 0).all(axis=2).mean() > 0.1                       # a visible black border: the BORDER_CONSTANT taps were exercised
    got2 = warp.warpRevBundle2(torch.from_numpy(img).to(cuda), torch.from_numpy(x_map).to(cuda), torch.from_numpy(y_map).to(cuda))
    assert np.array_equal(got2.cpu().numpy(), want)                          # (without the optional pixel-coordinate outputs)


@pytest.mark.parametrize("n,off", [(1, 0), (7, 0), (4096, 0), (720 * 1280, 0), (1001, 1), (4099, 3)])
def test_cvt_train2img_exact(cuda, n, off):
    """deploy_bundle.py:75: ((x + 0.5) * 255).astype(uint8) (clipped first); the vector path and the unaligned / tail path."""
    import torch
    from stabnet_amd import warp
    rng = np.random.default_rng(n)
    x = rng.uniform(-0.7, 0.7, size=n + off).astype(np.float32)
    x[: min(n, 4)] = [-0.5, 0.5, 0.49999997, -0.49803922][: min(n, 4)]
    want = ((x[off:] + np.float32(0.5)) * np.float32(255)).clip(0, 255).astype(np.uint8)
    xd = torch.from_numpy(x).to(cuda)[off:]
    buf = torch.zeros(n + off + 8, dtype=torch.uint8, device=cuda)
    out = warp.cvt_train2img(xd, buf[off:off + n])
    assert np.array_equal(out.cpu().numpy(), want)
    assert int(buf[off + n:].sum()) == 0 and int(buf[:off].sum()) == 0        # nothing written outside
