"""CPU: the decision plumbing of oracle/torch_ref.py (module docstring, DECISIONS).  Forcing an evaluation's OWN recorded
decisions must reproduce its value and gradient (to float64 summation order); forcing a flipped decision must change the gradient (so the
override really reaches the graph)."""
import numpy as np
import torch

from oracle import stabnet_oracle as O
from oracle import torch_ref as T


def _setup():
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    N, H, W = 1, 32, 64
    cfg = Config(height=H, width=W, batch_size=N, max_matches=16)
    ocfg = O.Config(height=H, width=W, batch_size=N, max_matches=16)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.3)
    b = synthetic.make_train_batch(cfg, N, H, W, 3)
    return ocfg, P, b


def _grad(P, b, ocfg, **kw):
    pt = {k: T.t(v, requires_grad=True) for k, v in P.items()}
    total, _ = T.train_objective(pt, b, ocfg, 1.0, 1.0, 0.0, training=True, **kw)
    total.backward()
    g = np.concatenate([(pt[k].grad.numpy() if pt[k].grad is not None else np.zeros(P[k].shape)).ravel() for k in sorted(P)])
    return float(total), g


def test_own_decisions_reproduce_the_gradient_and_a_flip_changes_it():
    torch.set_num_threads(4)
    ocfg, P, b = _setup()
    rec = {}
    tot0, g0 = _grad(P, b, ocfg, record=rec)
    assert set(rec) == {"1", "2"}
    for k in ("1", "2"):
        assert {"relu", "pool_argmax", "black", "corners"} <= set(rec[k])
        assert len(rec[k]["relu"]) == 49 + 3 and rec[k]["pool_argmax"].max() <= 8      # 48 unit BNs + postnorm, fc1..3
    tot1, g1 = _grad(P, b, ocfg, decisions=rec)
    # (not bit-identical: the overrides change tensor layouts and with them float64 summation orders; at this toy size block4's
    #  batch statistics are taken over TWO positions, which amplifies 1e-16 to ~1e-6 of the gradient scale)
    assert abs(tot1 - tot0) <= 1e-7 * abs(tot0) and np.abs(g1 - g0).max() <= 1e-4 * np.abs(g0).max()
    # flip a block of ReLU decisions of one late layer of tower 1: value and gradient must move
    key = "resnet_v2_50/block4/unit_3/bottleneck_v2/conv2/BatchNorm"
    flipped = {k: dict(v) for k, v in rec.items()}
    flipped["1"]["relu"] = dict(rec["1"]["relu"])
    flipped["1"]["relu"][key] = ~rec["1"]["relu"][key]
    tot2, g2 = _grad(P, b, ocfg, decisions=flipped)
    assert np.linalg.norm(g2 - g0) > 1e-3 * np.linalg.norm(g0)
    # the arg-max override reads the element it names: moving every arg-max to tap 4 (the window centre, always in frame) changes the pool
    moved = {k: dict(v) for k, v in rec.items()}
    moved["2"]["pool_argmax"] = np.full_like(rec["2"]["pool_argmax"], 4)
    tot3, _ = _grad(P, b, ocfg, decisions=moved)
    assert tot3 != tot0
