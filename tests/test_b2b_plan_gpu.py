"""GPU: the inference plan with conv2 -> conv3 of the block-1 / block-2 units fused into ONE launch (conv_b2b_kernel.h; a plan
switch that is OFF by default because it measured slower in the 720p frame -- DESIGN.md section 4, round 4).  The switch is read
once per process, so each plan runs in a fresh child.  theta of the fused plan against the ORACLE's regressor (the bar of every
regressor test: 2e-5) and against the default plan (the same products; where the default plan splits K the summation order
differs, nothing else)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import stabnet_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, tag, env_extra, shape):
    out = str(tmp_path / (tag + ".npz"))
    env = dict(os.environ, PYTHONPATH=ROOT, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "b2b_plan_child.py"), out, *[str(v) for v in shape]], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


@pytest.mark.parametrize("shape", [(1, 96, 160), (2, 72, 136)])      # 15 / 6 and 2 x (9 / 3) tiles in block 1 / 2; ragged last tiles
def test_fused_plan_matches_oracle_and_default_plan(cuda, tmp_path, shape):
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    N, H, W = shape
    fused = _run(tmp_path, "fused", {"STABNET_CONV_B2B_PLAN": "1", "STABNET_CONV_B2B_MIN_TILES": "1"}, shape)
    plain = _run(tmp_path, "plain", {"STABNET_CONV_B2B_PLAN": "0"}, shape)
    assert int(fused["launches"]) <= int(plain["launches"]) - 7          # one launch instead of two for each of the 7 units
    assert np.array_equal(fused["theta"], fused["theta2"])               # the same bits every time
    cfg = Config(height=H, width=W)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.3)
    want, _, _ = O.get_resnet(fused["x"], P, O.Config(height=H, width=W))
    assert np.abs(fused["theta"] - want).max() <= 2e-5
    assert np.abs(fused["theta"] - plain["theta"]).max() <= 5e-6
