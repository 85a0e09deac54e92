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
