"""GPU parity of the regressor and of the deploy step in conv operand mode 4 (packed split kernels: float32 operands as exact
sums of three bf16 terms, six partial products per product on v_mfma_f32_32x32x16_bf16, float32 accumulation) -- the mode
bench.py and deploy_bundle.py run by default.  The bars are those of the exact-f32-MFMA path (tests/test_regressor_gpu.py,
tests/test_baseline_sizes_gpu.py): theta within 2e-5 of the oracle, maps within 1e-4, black mask equal off the frame edge,
pixels within the Lipschitz bound -- plus theta within 2e-6 of the f32-MFMA path itself."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import stabnet_oracle as O
from test_baseline_sizes_gpu import _lipschitz_pixel_check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (1, 256, 256), (1, 288, 512)])
def test_regressor_mode4_matches_oracle(cuda, N, H, W):
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    from stabnet_amd.regressor import Regressor
    cfg, ocfg = Config(height=H, width=W), O.Config(height=H, width=W)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
    x, _ = synthetic.make_stack(cfg, N, H, W, seed=3)
    xt = torch.from_numpy(x).to(cuda)
    th0 = Regressor(P, N, H, W, cfg)(xt).cpu().numpy()
    th4 = Regressor(P, N, H, W, cfg, operand_mode=4)(xt).cpu().numpy()
    ref, _, _ = O.get_resnet(x, P, ocfg)
    d0, d4, d04 = np.abs(th0 - ref).max(), np.abs(th4 - ref).max(), np.abs(th4 - th0).max()
    print("theta max err vs oracle: f32 MFMA %.2e, packed split %.2e; between them %.2e (theta scale %.3f)" % (d0, d4, d04, np.abs(ref).max()))
    assert d4 <= 2e-5 and d0 <= 2e-5
    assert 0 < d04 <= 2e-6                    # another summation order (so not bit-equal), the same precision


def test_mode4_runs_the_packed_kernels(cuda):
    """The mode is really on: the frame's convolutions are conv_ring_f32_kernel<MODE, 4, KG, PRO> launches."""
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    from stabnet_amd.deploy import Profiler, StabNetStream
    H, W = 288, 512
    cfg = Config(height=H, width=W)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
    clip = torch.from_numpy(synthetic.make_clip(H, W, 3, seed=1234)).to(cuda)
    s = StabNetStream(P, H, W, cfg, streams=1, device=cuda, operand_mode=4)
    s.start(clip[0:1])
    prof = Profiler(400)
    s.step(clip[1:2], prof)
    names = [r[0] for r in prof.records()]
    convs = [n for n in names if n.startswith("conv_")]
    packed = [n for n in convs if n.startswith("conv_ring_f32_kernel<") and n.split(",")[1].strip() in ("4", "5")]
    flops = {True: 0.0, False: 0.0}
    for r in prof.records():
        if r[0].startswith("conv_") and r[2] > 0:
            flops[r[0] in packed] += r[2]
    assert len(packed) >= 40, (len(packed), sorted(set(convs)))
    assert flops[True] >= 0.9 * (flops[True] + flops[False])       # (low-K layers may stay on the register-staged f32 kernel)


@pytest.mark.parametrize("H,W,frames", [(720, 1280, (1, 2)), (1080, 1920, (1,))])
def test_deploy_step_mode4_matches_oracle_at_size(cuda, H, W, frames):
    """BASELINE configs[1] (1280x720) and the per-GPU shape of configs[4] (1920x1080) in mode 4, against the oracle."""
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    from stabnet_amd.deploy import StabNetStream
    cfg, ocfg = Config(height=H, width=W), O.Config(height=H, width=W)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
    clip = synthetic.make_clip(H, W, 3, seed=1234)
    s = StabNetStream(P, H, W, cfg, streams=1, device=cuda, operand_mode=4)
    s.start(torch.from_numpy(clip[0:1]).to(cuda))
    ring = O.DeployRing(clip[0], ocfg)
    for t in frames:
        got = s.step(torch.from_numpy(clip[t:t + 1]).to(cuda))
        torch.cuda.synchronize()
        ref, frame = O.deploy_step(ring, clip[t], P, ocfg)
        th = got["theta"].cpu().numpy()
        assert np.abs(th - ref["theta"]).max() <= 2e-5, (t, float(np.abs(th - ref["theta"]).max()))
        xm, ym = got["x_map"].cpu().numpy(), got["y_map"].cpu().numpy()
        assert np.abs(xm - ref["x_map"]).max() < 1e-4 and np.abs(ym - ref["y_map"]).max() < 1e-4, t
        flips = got["black_pix"].cpu().numpy() != ref["black_pix"]
        edge = (np.abs(np.abs(ref["x_map"][..., 0]) - 1) < 1e-4) | (np.abs(np.abs(ref["y_map"][..., 0]) - 1) < 1e-4)
        assert not (flips & ~edge).any(), t
        _lipschitz_pixel_check(clip[t], got["output"].cpu().numpy(), ref, xm, ym, H, W)
        d = np.abs(got["frame"].cpu().numpy()[0] - frame)
        assert np.quantile(d, 0.999) < 5e-3, (t, float(d.max()))


def test_mode4_graph_replay_equals_eager(cuda):
    """One hipGraph per stream replays the mode-4 frame bit-identically to eager launches, across a ring wrap."""
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    from stabnet_amd.deploy import StabNetStream
    H, W = 64, 96
    cfg = Config(height=H, width=W)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
    clip = torch.from_numpy(synthetic.make_clip(H, W, 40, seed=7)).to(cuda)
    a = StabNetStream(P, H, W, cfg, streams=1, device=cuda, operand_mode=4, use_graph=False)
    b = StabNetStream(P, H, W, cfg, streams=1, device=cuda, operand_mode=4, use_graph=True)
    a.start(clip[0:1]); b.start(clip[0:1])
    for t in range(1, 40):
        ra = {k: v.clone() for k, v in a.step(clip[t:t + 1]).items()}
        rb = b.step(clip[t:t + 1])
        for k in ra:
            assert torch.equal(ra[k], rb[k]), (t, k)
    assert b._graph is not None
