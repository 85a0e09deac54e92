"""GPU: stabnet_adam_step against the oracle's restatement of TF 1.3's ApplyAdam + AdamOptimizer bookkeeping
(train_bundle_nobm.py:155-160), driven with FIXED gradients over several steps so that the
epsilon placement, the bias correction (float32 beta-power variables) and the m/v update form are all exercised."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import stabnet_oracle as O

pytestmark = pytest.mark.gpu


def _grads(rng, n, steps):
    # magnitudes from 1e-9 (below epsilon: the eps placement matters) to 1, a few exact zeros, sign changes between steps
    out = []
    for _ in range(steps):
        g = (rng.standard_normal(n) * 10.0 ** rng.integers(-9, 1, n)).astype(np.float32)
        g[rng.random(n) < 0.02] = 0.0
        out.append(g)
    return out


@pytest.mark.parametrize("n", [4096, 1001])                # 1001: the scalar tail path
def test_adam_five_steps_match_tf_restatement(cuda, n):
    from stabnet_amd import _lib
    rng = np.random.default_rng(n)
    w0 = rng.standard_normal(n).astype(np.float32) * 0.1
    gs = _grads(rng, n, 5)
    a32, a64 = O.AdamTF(n), O.AdamTF(n, dtype=np.float64)
    w32, w64 = w0.copy(), w0.astype(np.float64)
    w = torch.from_numpy(w0.copy()).to(cuda)
    m = torch.zeros(n, device=cuda)
    v = torch.zeros(n, device=cuda)
    st = torch.cuda.current_stream().cuda_stream
    for t, g in enumerate(gs, start=1):
        lr = float(O.exponential_decay_staircase(2e-5, t - 1, 40000, 0.1))
        w32 = a32.step(w32, g, lr)
        w64 = a64.step(w64, g, lr)
        gd = torch.from_numpy(g).to(cuda)
        _lib.call("stabnet_adam_step", w.data_ptr(), gd.data_ptr(), 0, m.data_ptr(), v.data_ptr(), n, lr, 0.9, 0.999, 1e-8,
                  t, 1.0, st)
        torch.cuda.synchronize()
        # same float32 operations in the same order: the moments are bit-identical, the weights to 1 ulp (division/sqrt
        # are correctly rounded on both sides; kept at 1 ulp in case a libm sqrt differs in the last place)
        assert np.array_equal(m.cpu().numpy(), a32.m), t
        assert np.array_equal(v.cpu().numpy(), a32.v), t
        got = w.cpu().numpy()
        ulp = np.spacing(np.abs(w32)).astype(np.float32)
        assert (np.abs(got - w32) <= ulp).all(), (t, float(np.abs(got - w32).max()))
        # exact-arithmetic shadow: the UPDATE (not the weight, whose float32 storage rounds at 6e-8 relative) to 1e-6
        upd, upd64 = got.astype(np.float64) - w0, w64 - w0
        assert np.abs(upd - upd64).max() <= 1e-6 * np.abs(upd64).max() + 2 * np.spacing(np.float32(np.abs(w0).max())) * t, t
    # step 1 of TF Adam moves every element with |g| >> eps by lr (bias-corrected m / sqrt(v) = sign(g)); elements with
    # |g| << eps move by ~ lr * g / eps: this is what tells TF's eps placement from "epsilon-hat" variants
    a = O.AdamTF(3)
    g = np.array([1e-3, 1e-9, -1e-12], np.float32)
    w1 = a.step(np.zeros(3, np.float32), g, 2e-5)
    # closed form of step 1: m = 0.1 g, v = 0.001 g^2, alpha = lr * sqrt(0.001) / 0.1  ->  dw = -lr g / (|g| + eps / sqrt(0.001))
    eps_hat = 1e-8 / np.sqrt(1e-3)
    want = -2e-5 * g.astype(np.float64) / (np.abs(g.astype(np.float64)) + eps_hat)
    assert np.allclose(w1, want, rtol=1e-4, atol=0)

def test_adam_two_gradient_buffers_and_scale(cuda):
    """g = (grads + grads2) * gscale: the data-parallel form (sum of the ranks' gradients divided by world)."""
    from stabnet_amd import _lib
    n = 2048
    rng = np.random.default_rng(7)
    w0 = rng.standard_normal(n).astype(np.float32)
    g1, g2 = _grads(rng, n, 2)
    a = O.AdamTF(n)
    want = a.step(w0, ((g1 + g2) * np.float32(0.5)).astype(np.float32), 2e-5)
    w = torch.from_numpy(w0.copy()).to(cuda)
    m = torch.zeros(n, device=cuda); v = torch.zeros(n, device=cuda)
    d1, d2 = torch.from_numpy(g1).to(cuda), torch.from_numpy(g2).to(cuda)       # (kept alive across the call)
    _lib.call("stabnet_adam_step", w.data_ptr(), d1.data_ptr(), d2.data_ptr(), m.data_ptr(), v.data_ptr(), n, 2e-5, 0.9, 0.999,
              1e-8, 1, 0.5, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(m.cpu().numpy(), a.m)
    assert (np.abs(w.cpu().numpy() - want) <= np.spacing(np.abs(want))).all()
