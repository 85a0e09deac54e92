"""GPU: the RCCL code path on the one card the box has.  ONE fresh child rank with backend "nccl" runs Trainer with the
communication path forced on -- init_process_group("nccl", device_id=...), every gradient bucket all-reduced as a slice of
the ONE gradient buffer on the communication stream, the wait_stream joins before weight decay / Adam (train.py) -- and
the result must equal the run without a process group BIT FOR BIT (a sum over one rank is the identity; any missing
stream ordering shows up as a difference).  No reference counterpart: train_bundle_nobm.py:199-201 is single-device."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_one_rank_rccl_trainer_equals_no_group(cuda, tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py"), str(tmp_path)], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = np.load(tmp_path / "rccl.npz")
    assert bool(r["rccl_mapped"]) and bool(r["probe_ok"])
    steps = 3
    # 4 stage buckets + the BN gamma/beta range per step, together exactly the trainables
    assert int(r["n_collectives"]) == 5 * steps
    assert int(r["bucket_bytes"][:5].sum()) == 4 * int(r["nt"])
    assert np.isfinite(r["allreduce_ms"]).all() and (r["allreduce_ms"] >= 0).all()
    for k in ("params", "grads", "m"):
        a, b = r["plain_" + k], r["rccl_" + k]
        assert np.isfinite(a).all()
        assert np.array_equal(a, b), "%s differ between the no-group and the 1-rank RCCL run: max %g" % (k, np.abs(a - b).max())
