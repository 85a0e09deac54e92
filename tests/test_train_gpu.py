"""GPU: one full siamese training step (two towers, batch-stat BN, warp, all losses, temporal loss, backward, weight
decay, Adam) against the torch float64 autograd oracle of the reference objective (train_bundle_nobm.py:107-160)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import stabnet_oracle as O
from oracle import torch_ref as T

pytestmark = pytest.mark.gpu


def _setup(N, H, W, seed=5):
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    cfg = Config(height=H, width=W, batch_size=N, max_matches=48)
    ocfg = O.Config(height=H, width=W, batch_size=N, max_matches=48)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.3)
    b = synthetic.make_train_batch(cfg, N, H, W, seed)
    b["flow"] = (b["flow"] + np.random.default_rng(1).normal(0, 0.02, b["flow"].shape)).astype(np.float32)
    return cfg, ocfg, P, b


def test_training_step_matches_autograd_oracle(cuda):
    from stabnet_amd.train import Trainer
    N, H, W = 2, 64, 96
    cfg, ocfg, P, b = _setup(N, H, W)
    gates = {"use_theta_loss": 1, "use_temp_loss": 1, "use_black_loss": 1, "use_theta_only": 0}

    pt = {k: T.t(v, requires_grad=True) for k, v in P.items()}
    stats, own = {}, {}
    total, parts = T.train_objective(pt, b, ocfg, 1.0, 1.0, 0.0, training=True, batch_stats=stats, record=own)
    total.backward()
    want_g = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in pt.items()}

    tr = Trainer(P, N, H, W, cfg, device=cuda)
    p0 = tr.params.clone()
    dev_b = {k: torch.from_numpy(v).to(cuda) for k, v in b.items()}
    tr.forward_backward(dev_b, gates, apply_update=False)
    torch.cuda.synchronize()

    # forward: theta of both towers and the loss terms
    for k, key in enumerate(("tower1", "tower2")):
        assert np.abs(tr.theta[k].cpu().numpy() - parts[key]["theta"].detach().numpy()).max() < 5e-5
    lo = tr.losses()
    assert lo["total_loss"] == pytest.approx(float(total), rel=2e-3)
    assert lo["temp_loss"] == pytest.approx(float(parts["temp_loss"]) * cfg.temp_mul, rel=5e-3, abs=1e-6)
    assert lo["tower1"]["img_loss"] == pytest.approx(float(parts["tower1"]["img"]) * cfg.img_mul, rel=2e-3)
    assert lo["tower2"]["feature_loss"] == pytest.approx(float(parts["tower2"]["feature"]), rel=2e-3)

    # backward: every trainable tensor, error relative to that tensor's gradient scale.
    # The objective is piecewise smooth (ReLU signs, max-pool arg-max, `black_pix` = a strict comparison of the warp map with +-1,
    # the sampler's floor corners): a float32 forward whose pre-activations differ from the float64 ones in the 7th digit can
    # take the other side of such a decision, and then the two sides differentiate DIFFERENT smooth pieces.  So the gradient is
    # checked twice: (a) against the float64 autograd of the piece the GPU forward really took -- its decisions are read back
    # from the training workspace and forced onto the oracle (tests/_decisions.py, oracle/torch_ref.py DECISIONS) -- with the
    # tight per-element bar; (b) against the un-forced float64 gradient with the whole-gradient bars (a flipped decision moves
    # single elements by several per cent of their tensor's scale, the whole gradient by ~1e-3).
    from _decisions import flips, gpu_decisions, gradient_errors
    got_flat = tr.grad_flat().cpu().numpy()
    want_flat = tr.plan.pack({k: want_g[k] for k in P})[:tr.nt]
    dec = gpu_decisions(tr)
    flipped = flips(dec, own)
    print("DECISIONS that differ between the float32 forward and float64:", flipped)
    assert sum(n for _, _, n, _ in flipped) <= 200, flipped          # a handful of rounding-level flips, not a systematic difference
    pf = {k: T.t(v, requires_grad=True) for k, v in P.items()}
    total_f, _ = T.train_objective(pf, b, ocfg, 1.0, 1.0, 0.0, training=True, decisions=dec)
    total_f.backward()
    forced_flat = tr.plan.pack({k: (pf[k].grad.numpy() if pf[k].grad is not None else np.zeros(P[k].shape)) for k in P})[:tr.nt]
    worst_abs, worst_l2, whole, rows = gradient_errors(tr.plan, got_flat, forced_flat)
    print("MEASURED (decisions forced) worst element %.3e worst tensor L2 %.3e whole L2 %.3e" % (worst_abs, worst_l2, whole))
    # measured: ONE decision differs at this size -- a ReLU sign in tower 2, block3/unit_2/preact (1 of 49 152) --; with it forced:
    # worst element 1.3e-2, worst tensor 4.0e-3, whole gradient 8.3e-6 (un-forced: 8.4e-2 / 8.7e-3 / 2.4e-3)
    for name, err, l2 in rows:
        assert err < 2e-2, "%s: element err %g with the forward's decisions forced" % (name, err)
        assert l2 < 6e-3, "%s: relative L2 err %g with the forward's decisions forced" % (name, l2)
    assert whole < 1e-4, whole
    # (b) un-forced: whole-gradient agreement
    u_abs, u_l2, whole_u, _ = gradient_errors(tr.plan, got_flat, want_flat)
    print("MEASURED (un-forced) worst element %.3e worst tensor L2 %.3e whole L2 %.3e" % (u_abs, u_l2, whole_u))
    assert whole_u < 5e-3, whole_u                                                      # measured 1.7e-4 .. 2.4e-3 (a flip or not)
    cos = float(np.dot(got_flat, want_flat) / (np.linalg.norm(got_flat) * np.linalg.norm(want_flat)))
    assert cos > 1 - 1e-5, "gradient cosine %r" % cos

    # batch-statistics BN moving averages (decay 0.997), both towers applied
    q = tr.plan.unpack(tr.params.cpu().numpy())
    name = "resnet_v2_50/block1/unit_1/bottleneck_v2/preact/moving_mean"
    m = P[name].astype(np.float64)
    for key in ("1", "2"):
        m = m - (m - stats[key][name[:-len("/moving_mean")]][0]) * (1 - cfg.bn_decay)
    assert np.abs(q[name] - m).max() < 1e-5
    assert torch.equal(tr.params[:tr.nt], p0[:tr.nt])            # apply_update=False left the trainables alone

    # the step is reproducible bit for bit: a second trainer gives the same gradient, and the applied update is exactly
    # what the oracle's restatement of TF Adam makes of that gradient (train_bundle_nobm.py:155-160)
    tr2 = Trainer(P, N, H, W, cfg, device=cuda)
    tr2.forward_backward(dev_b, gates, apply_update=True)
    torch.cuda.synchronize()
    assert np.array_equal(tr2.grad_flat().cpu().numpy(), got_flat)
    adam = O.AdamTF(tr.nt)
    want_w = adam.step(p0[:tr.nt].cpu().numpy(), got_flat, float(O.exponential_decay_staircase(cfg.initial_learning_rate, 0, cfg.step_size, 0.1)))
    got_w = tr2.params[:tr2.nt].cpu().numpy()
    assert (np.abs(got_w - want_w) <= np.spacing(np.abs(want_w))).all()
    assert np.array_equal(tr2.adam_m.cpu().numpy(), adam.m) and np.array_equal(tr2.adam_v.cpu().numpy(), adam.v)


def test_theta_only_phase_and_gates(cuda):
    """i <= do_theta_only_iter: total = theta_loss only (s_net_bundle_nobm.py:357-359)."""
    from stabnet_amd.train import Trainer, loss_gates, learning_rate
    N, H, W = 2, 64, 96
    cfg, ocfg, P, b = _setup(N, H, W)
    assert loss_gates(0, cfg) == {"use_theta_loss": 1, "use_temp_loss": 0, "use_black_loss": 0, "use_theta_only": 1}
    assert loss_gates(5000, cfg) == {"use_theta_loss": 1, "use_temp_loss": 1, "use_black_loss": 1, "use_theta_only": 0}
    assert learning_rate(39999, cfg) == pytest.approx(2e-5) and learning_rate(40000, cfg) == pytest.approx(2e-6)
    tr = Trainer(P, N, H, W, cfg, device=cuda)
    dev_b = {k: torch.from_numpy(v).to(cuda) for k, v in b.items()}
    tr.forward_backward(dev_b, loss_gates(0, cfg), apply_update=False)
    lo = tr.losses()
    assert lo["total_loss"] == pytest.approx(lo["tower1"]["theta_loss"] + lo["tower2"]["theta_loss"], rel=1e-6)
    # only the path to theta has gradient: conv1 weights do, and there is no weight-decay term
    pt = {k: T.t(v, requires_grad=True) for k, v in P.items()}
    total, _ = T.train_objective(pt, b, ocfg, 0.0, 0.0, 1.0, training=True)
    total.backward()
    want = tr.plan.pack({k: (pt[k].grad.numpy() if pt[k].grad is not None else np.zeros(P[k].shape)) for k in P})[:tr.nt]
    got = tr.grad_flat().cpu().numpy()
    cos = float(np.dot(got, want) / (np.linalg.norm(got) * np.linalg.norm(want)))
    assert cos > 1 - 1e-5


def test_loss_decreases_on_a_fixed_batch(cuda):
    """End-to-end sanity of the optimiser loop: 25 Adam steps on one fixed batch reduce the total loss, reproducibly.
    The learning rate is the reference's 2e-5 (configs/v2_93.py:6).  An earlier version of this test used 2e-4 to make the
    decrease faster; that is 10x the reference's step on a TWO-sample batch whose batch-norm statistics move with every
    update, and the trajectory went 15.8 -> 8.1 -> 19.6 (not a bug of the step: the same happens in float64).  At the
    reference's rate the loss falls monotonically."""
    from stabnet_amd.config import Config
    from stabnet_amd.train import Trainer
    from stabnet_amd import synthetic
    N, H, W = 2, 64, 96
    cfg = Config(height=H, width=W, batch_size=N, max_matches=48)         # the reference's learning rate, 2e-5
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.3)
    b = synthetic.make_train_batch(cfg, N, H, W, 5)
    dev_b = {k: torch.from_numpy(v).to(cuda) for k, v in b.items()}
    gates = {"use_theta_loss": 1, "use_temp_loss": 1, "use_black_loss": 1, "use_theta_only": 0}
    tr = Trainer(P, N, H, W, cfg, device=cuda)
    losses = []
    for _ in range(25):
        tr.forward_backward(dev_b, gates)
        losses.append(tr.losses()["total_loss"])
    assert all(np.isfinite(losses))
    # a fixed trajectory (the step is deterministic): strictly decreasing over the 25 steps, to about half its start
    # (measured 15.78 -> 7.98)
    assert all(b < a for a, b in zip(losses, losses[1:])), losses
    assert losses[-1] < 0.55 * losses[0], losses[::3]
    # ... and a second run reproduces it exactly
    tr_b = Trainer(P, N, H, W, cfg, device=cuda)
    losses_b = []
    for _ in range(25):
        tr_b.forward_backward(dev_b, gates)
        losses_b.append(tr_b.losses()["total_loss"])
    assert losses_b == losses
    assert torch.equal(tr_b.params, tr.params)
    # checkpoint round trip restores the optimiser state exactly: the next step is bit-identical
    sd = tr.state_dict()
    tr2 = Trainer(P, N, H, W, cfg, device=cuda)
    tr2.load_state_dict(sd)
    tr.forward_backward(dev_b, gates)
    tr2.forward_backward(dev_b, gates)
    assert torch.equal(tr.params, tr2.params) and torch.equal(tr.adam_m, tr2.adam_m) and torch.equal(tr.adam_v, tr2.adam_v)
    assert tr.global_step == tr2.global_step == 26
