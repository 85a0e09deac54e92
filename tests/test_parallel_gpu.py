"""GPU: the data-parallel Trainer with world_size 2.  The test process itself does not start the ranks in-process: it
spawns two fresh children (torch.distributed.run, gloo backend, both on cuda:0 -- the box has one GPU; on an 8-GPU node
the same code runs over RCCL) and checks what SURVEY 8e specifies: ONE summed gradient buffer, buckets in reverse layer
order, identical parameters on every rank, update = Adam on the MEAN of the ranks' single-process gradients (local BN)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import stabnet_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_trainer_step(cuda, tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", STABNET_TEST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_child.py"), str(tmp_path)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in (0, 1)]
    world = 2
    # (1) one exchange of ONE gradient buffer: 4 stage buckets + the BN gamma/beta range, together exactly the trainables
    nt = r[0]["g_dp"].size
    assert int(r[0]["n_buckets"]) == 5 and int(r[0]["bucket_bytes"].sum()) == 4 * nt
    # (2) both ranks hold the same summed gradient and the same parameters / optimiser state after the step
    assert np.array_equal(r[0]["g_dp"], r[1]["g_dp"]) and np.array_equal(r[0]["p1"], r[1]["p1"])
    assert np.array_equal(r[0]["adam_m"], r[1]["adam_m"])
    # (3) the summed gradient = sum of the ranks' single-process gradients (each includes the L2 term once; the DP step adds
    #     it `world` times to the sum, so sum/world = mean of the single-process gradients): float32 sum-order tolerance
    mean_single = (r[0]["g_single"].astype(np.float64) + r[1]["g_single"].astype(np.float64)) / world
    g_eff = r[0]["g_dp"].astype(np.float64) / world
    gmax = np.abs(mean_single).max()
    assert np.abs(g_eff - mean_single).max() <= 1e-5 * gmax
    assert not np.allclose(r[0]["g_single"], r[1]["g_single"], rtol=1e-3, atol=1e-6 * gmax)     # the shards do differ
    # (4) the applied update is TF Adam on that mean (gscale = 1/world inside the kernel)
    adam = O.AdamTF(nt)
    want = adam.step(r[0]["p0"], (r[0]["g_dp"] * np.float32(1.0 / world)).astype(np.float32), 2e-5)
    assert (np.abs(r[0]["p1"] - want) <= np.spacing(np.abs(want))).all()
    assert np.array_equal(r[0]["adam_m"], adam.m)
    # (5) BN moving statistics are per-rank (local BN) until sync_moving_stats() averages them for the checkpoint
    assert not np.array_equal(r[0]["mov_before"], r[1]["mov_before"])
    assert np.array_equal(r[0]["mov_after"], r[1]["mov_after"])
    assert np.allclose(r[0]["mov_after"], (r[0]["mov_before"] + r[1]["mov_before"]) / 2, rtol=1e-6, atol=1e-7)
