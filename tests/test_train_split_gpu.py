"""GPU: the training step with its weight-operand launches on the packed split kernels (Trainer(split_operands=True); opt-in:
float32 operands as exact sums of three bf16 terms on the bf16 matrix pipe, an image of the re-packed dgrad weights written once
per step, and its prologue-carrying 1x1 forward pairs likewise) against the same step on the exact-f32-MFMA kernels
(split_operands=False): float32 summation order is all that differs."""
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
    assert n0 == 0 and n1 >= 60, (n0, n1)                       # the mode is really on: the stride-1 dgrad launches and the 1x1 forward pairs
    assert l1["total_loss"] == pytest.approx(l0["total_loss"], rel=1e-5)
    # per parameter tensor: difference relative to the tensor's own gradient scale.  The forward's activations move in the 7th digit,
    # so a ReLU / arg-max decision can fall the other way (tests/test_train_gpu.py): single elements may then differ by per cents of
    # their tensor's scale, whole tensors by ~1e-3 -- the bars of the un-forced comparison there.
    worst_el, worst_l2 = 0.0, 0.0
    for name, off, kind, dims, aux in plan.table:
        n = int(np.prod(dims))
        if n == 0 or off + n > g0.size:
            continue
        a, c = g0[off:off + n].astype(np.float64), g1[off:off + n].astype(np.float64)
        sc = np.abs(a).max()
        if sc > 0:
            worst_el = max(worst_el, float(np.abs(a - c).max() / sc))
            worst_l2 = max(worst_l2, float(np.linalg.norm(a - c) / max(np.linalg.norm(a), 1e-30)))
    whole = float(np.linalg.norm(g0.astype(np.float64) - g1) / np.linalg.norm(g0.astype(np.float64)))
    print("split step vs f32 MFMA step: worst element %.3e, worst tensor L2 %.3e, whole gradient L2 %.3e" % (worst_el, worst_l2, whole))
    assert worst_el < 1e-1 and worst_l2 < 2e-2 and whole < 5e-3
    # and the run is reproducible bit for bit
    tr = Trainer(P, N, H, W, cfg, device=cuda, split_operands=True)
    tr.forward_backward(b, gates, apply_update=False)
    torch.cuda.synchronize()
    assert np.array_equal(tr.grad_flat().cpu().numpy(), g1)
