"""GPU: the training backward with its dgrad launches on the packed split kernels (Trainer(split_operands=True); opt-in:
float32 operands as exact sums of three bf16 terms on the bf16 matrix pipe, an image of the re-packed dgrad weights written once
per step) against the same step on the exact-f32-MFMA kernels (split_operands=False).  The forward is the same code, so the
losses are bit-identical; the gradients differ by float32 summation order only."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (8, 288, 512)])
def test_split_dgrad_step_equals_f32_mfma_step(cuda, N, H, W):
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    from stabnet_amd.deploy import Profiler
    from stabnet_amd.train import Trainer
    cfg = Config(height=H, width=W, batch_size=N, max_matches=48)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.3)
    b = {k: torch.from_numpy(v).to(cuda) for k, v in synthetic.make_train_batch(cfg, N, H, W, 5).items()}
    gates = {"use_theta_loss": 1, "use_temp_loss": 1, "use_black_loss": 1, "use_theta_only": 0}
    res = {}
    for split in (False, True):
        tr = Trainer(P, N, H, W, cfg, device=cuda, split_operands=split)
        assert tr.split_operands == split
        prof = Profiler(4000)
        tr.prof = prof
        tr.forward_backward(b, gates, apply_update=False)
        torch.cuda.synchronize()
        tr.prof = None
        names = [r[0] for r in prof.records()]
        packed = [n for n in names if n.startswith("conv_ring_f32_kernel<") and n.split(",")[1].strip() in ("4", "5")]
        res[split] = (tr.grad_flat().cpu().numpy().copy(), tr.losses(), tr.plan, len(packed))
        del tr
        torch.cuda.empty_cache()
    g0, l0, plan, n0 = res[False]
    g1, l1, _, n1 = res[True]
    assert n0 == 0 and n1 >= 30, (n0, n1)                       # the mode is really on: the stride-1 dgrad launches of the 16 units
    assert l0["total_loss"] == l1["total_loss"]                  # same forward
    # per parameter tensor: difference relative to the tensor's own gradient scale
    worst = 0.0
    for name, off, kind, dims, aux in plan.table:
        n = int(np.prod(dims))
        if n == 0 or off + n > g0.size:
            continue
        a, c = g0[off:off + n], g1[off:off + n]
        s = np.abs(a).max()
        if s > 0:
            worst = max(worst, float(np.abs(a - c).max() / s))
    print("worst per-tensor relative gradient difference, split dgrad vs f32 MFMA dgrad: %.3e" % worst)
    assert worst < 2e-4
    # and the run is reproducible bit for bit
    tr = Trainer(P, N, H, W, cfg, device=cuda, split_operands=True)
    tr.forward_backward(b, gates, apply_update=False)
    torch.cuda.synchronize()
    assert np.array_equal(tr.grad_flat().cpu().numpy(), g1)
