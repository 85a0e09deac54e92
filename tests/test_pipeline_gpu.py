"""ClipPipeline (upload / frame / download on three streams) against the serial loop of the driver: same bytes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _serial(stream, grey, bgr, warp):
    dev = stream.reg.device
    H, W = stream.H, stream.W
    stream.start(torch.from_numpy(grey[0][None]).to(dev))
    out = []
    for t in range(1, len(grey)):
        r = stream.step(torch.from_numpy(grey[t][None]).to(dev))
        o = ((r["output"][0, :, :, 0].cpu().numpy() + 0.5) * 255).clip(0, 255).astype(np.uint8)
        c = warp.warpRevBundle2(torch.from_numpy(bgr[t]).to(dev), r["x_map"], r["y_map"]).cpu().numpy()
        out.append({"t": t, "output": o, "bgr": c, "x_map": r["x_map"].view(H, W).cpu().numpy(),
                    "y_map": r["y_map"].view(H, W).cpu().numpy(), "black": r["black_pix"].view(H, W).cpu().numpy().astype(np.uint8)})
    return out


@pytest.mark.parametrize("slots,maps", [(2, False), (3, True), (5, False)])
def test_pipeline_matches_serial_loop(cuda, slots, maps):
    from stabnet_amd import synthetic, warp
    from stabnet_amd.config import Config
    from stabnet_amd.deploy import ClipPipeline, StabNetStream
    H, W, T = 96, 128, 23
    cfg = Config(height=H, width=W)
    params = synthetic.make_params(cfg, seed=3, theta_scale=0.2)
    grey = synthetic.make_clip(H, W, T, seed=11).astype(np.float32)
    rng = np.random.default_rng(5)
    bgr = rng.integers(0, 256, size=(T, H, W, 3), dtype=np.uint8)
    want = _serial(StabNetStream(params, H, W, cfg, device=cuda, use_graph=True), grey, bgr, warp)
    pipe = ClipPipeline(StabNetStream(params, H, W, cfg, device=cuda, use_graph=True), colour=True, slots=slots)
    got = pipe.run(grey, bgr, maps=maps)
    assert [g["t"] for g in got] == list(range(1, T))
    for g, w in zip(got, want):
        assert np.array_equal(g["output"], w["output"]), g["t"]
        assert np.array_equal(g["bgr"], w["bgr"]), g["t"]
        if maps:
            for k in ("x_map", "y_map"):
                assert np.array_equal(g[k], w[k]), (g["t"], k)
            assert np.array_equal(g["black"], w["black"].astype(np.uint8)), g["t"]
    # a second clip through the same pipeline (slot events and the ring are re-armed) and the sink form
    seen = []
    pipe.run(grey[:9], bgr[:9], sink=lambda r: seen.append((r["t"], r["output"].copy(), r["bgr"].copy())), maps=maps)
    assert [s[0] for s in seen] == list(range(1, 9))
    for s, w in zip(seen, want):
        assert np.array_equal(s[1], w["output"]) and np.array_equal(s[2], w["bgr"])


def test_pipeline_argument_errors(cuda):
    from stabnet_amd import _lib, synthetic
    from stabnet_amd.config import Config
    from stabnet_amd.deploy import ClipPipeline, StabNetStream
    cfg = Config(height=64, width=96)
    st = StabNetStream(synthetic.make_params(cfg, seed=0), 64, 96, cfg, device=cuda)
    with pytest.raises(_lib.StabnetError):
        ClipPipeline(st, slots=1)
    with pytest.raises(_lib.StabnetError):
        ClipPipeline(st, colour=True).run(np.zeros((3, 64, 96), np.float32))
