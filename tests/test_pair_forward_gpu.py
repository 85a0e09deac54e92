"""GPU: the training forward's PAIR launches (conv_igemm_f32_pair_kernel: one launch convolves both siamese towers, the
workgroups of the second tower shift five tower-owned pointers -- input, output, residual, BN scale, BN shift) against one
launch per tower, EXACTLY.  Both forms run the same kernel body on the same 64 x 64 tiles; with the tile, split-K and kernel
family pinned by the debug switches (STABNET_CONV_TILE=2, STABNET_CONV_SPLITK=1, STABNET_CONV_RING=0) every output element is the
same sum in the same order, so every kept activation of both towers, every batch statistic, theta, the updated moving
averages and the whole gradient must be BIT-IDENTICAL.  A wrong pointer shift for tower 2 (dx, dy, dres, dscale, the
out-of-frame "safe" address) cannot hide behind a tolerance here.  The switches are read once per process, so each form runs in
a fresh child.  Shape: tower rows are a multiple of 64 in the stem, block 1 and block 2 (paired launches) and not in blocks
3 / 4 (per-tower launches inside the same lockstep forward)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, tag, pair):
    out = str(tmp_path / (tag + ".npz"))
    env = dict(os.environ, PYTHONPATH=ROOT, STABNET_TRAIN_PAIR_FWD=str(pair), STABNET_CONV_TILE="2", STABNET_CONV_SPLITK="1",
               STABNET_CONV_RING="0", STABNET_CONV_NBUF="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pairfwd_child.py"), out, "4", "96", "160"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


def test_pair_launches_equal_per_tower_launches_bit_for_bit(cuda, tmp_path):
    a, b = _run(tmp_path, "pair", 1), _run(tmp_path, "single", 0)
    assert np.isfinite(a["theta"].view(np.float32)).all() and np.abs(a["grads"].view(np.float32)).max() > 0
    for key in ("theta", "bn", "acts", "params", "grads"):
        x, y = a[key], b[key]
        assert x.shape == y.shape
        nd = int((x != y).sum())
        assert nd == 0, "%s: %d of %d words differ between the pair launches and one launch per tower (first at %s)" % (
            key, nd, x.size, np.argwhere(x != y)[:3].tolist())


def _run_env(tmp_path, tag, extra_env, shape=("4", "96", "160")):
    out = str(tmp_path / (tag + ".npz"))
    env = dict(os.environ, PYTHONPATH=ROOT, **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pairfwd_child.py"), out, *shape], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


def test_rowrun_stem_training_step_equals_the_padded_stem(cuda, tmp_path):
    """The training stem on the tight zero-bordered 13-channel operand (ring kernel forward, row-run wgrad + scatter) against
    the same step on the channel-padded stack (STABNET_STEM_ROWRUN=0: register-staged forward, general wgrad): the same
    products in another grouping, so theta, the batch statistics, the moving averages and the whole gradient agree to float32
    summation noise -- and the stem's own weight / bias gradient, which only the stem path produces, does too."""
    from stabnet_amd.config import Config
    from stabnet_amd.regressor import NetPlan
    a, b = _run_env(tmp_path, "rowrun", {}), _run_env(tmp_path, "padded", {"STABNET_STEM_ROWRUN": "0"})
    f = lambda z, k: z[k].view(np.float32).astype(np.float64)
    th_a, th_b = f(a, "theta"), f(b, "theta")
    assert np.isfinite(th_a).all() and np.abs(th_a - th_b).max() <= 2e-5 * max(np.abs(th_b).max(), 1.0)
    assert np.abs(f(a, "bn") - f(b, "bn")).max() <= 1e-4 * np.abs(f(b, "bn")).max()
    assert np.abs(f(a, "params") - f(b, "params")).max() <= 1e-5            # (moving averages updated by the forward)
    ga, gb = f(a, "grads"), f(b, "grads")
    assert np.linalg.norm(ga - gb) <= 5e-3 * np.linalg.norm(gb)
    plan = NetPlan(4, 96, 160, Config(height=96, width=160, batch_size=4), keep_activations=True)
    for name, off, kind, dims, aux in plan.table:
        if name in ("resnet_v2_50/conv1/weights", "resnet_v2_50/conv1/biases"):
            n = int(np.prod([d for d in dims if d > 0]))
            sa, sb = ga[off:off + n], gb[off:off + n]
            if name.endswith("weights"):
                w_scale = np.abs(sb).max()
                assert w_scale > 0
                assert np.linalg.norm(sa - sb) <= 5e-3 * np.linalg.norm(sb), name
                # pad channels 13..15 of OHWI [64][7][7][16] get nothing
                assert not sa.reshape(dims)[..., aux:].any() and not sb.reshape(dims)[..., aux:].any()
            else:
                # a bias in front of max-pool + batch norm has a gradient of exactly zero in exact arithmetic (BN removes the
                # shift): both forms leave float32 cancellation noise, far below the weights' gradient
                assert max(np.abs(sa).max(), np.abs(sb).max()) <= 1e-3 * w_scale, name
