"""GPU: the two siamese towers in lockstep (stabnet_towers_fwd_train / stabnet_towers_bwd_stage: grouped BN reductions, one
conv / wgrad / dgrad launch for both towers where the shapes allow it, FC head on the pair) against the plain composition the
reference graph describes -- tower 1, then tower 2, over the same weights (train_bundle_nobm.py:107-108)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (4, 96, 160)])
def test_lockstep_towers_match_tower_after_tower(cuda, N, H, W):
    from stabnet_amd import _lib, synthetic
    from stabnet_amd._tensor import ptr, stream_ptr
    from stabnet_amd.config import Config
    from stabnet_amd.regressor import NetPlan
    cfg = Config(height=H, width=W, batch_size=N)
    plan = NetPlan(N, H, W, cfg, keep_activations=True)
    flat = plan.pack(synthetic.make_params(cfg, seed=0, theta_scale=0.3))
    rng = np.random.default_rng(3)
    xs = [torch.from_numpy(rng.uniform(-0.5, 0.5, (N, H, W, cfg.in_ch)).astype(np.float32)).to(cuda) for _ in range(2)]
    dth = [torch.from_numpy(rng.standard_normal((N, cfg.n_theta)).astype(np.float32)).to(cuda) for _ in range(2)]
    L = _lib.lib()
    nb = L.stabnet_net_train_workspace_bytes(plan.handle)
    nt = plan.n_trainable
    st = stream_ptr(cuda)

    def run(lockstep):
        params = torch.from_numpy(flat.copy()).to(cuda)
        ws = [torch.zeros(nb, dtype=torch.uint8, device=cuda) for _ in range(2)]
        th = [torch.empty((N, cfg.n_theta), dtype=torch.float32, device=cuda) for _ in range(2)]
        grads = torch.zeros(nt, dtype=torch.float32, device=cuda)
        if lockstep:
            _lib.call("stabnet_towers_fwd_train", plan.handle, ptr(params), ptr(xs[0]), ptr(xs[1]), ptr(th[0]), ptr(th[1]), ptr(ws[0]),
                      ptr(ws[1]), nb, cfg.bn_eps, cfg.bn_decay, st, 0, device=cuda)
            for stage in range(L.stabnet_net_num_grad_stages()):
                _lib.call("stabnet_towers_bwd_stage", plan.handle, ptr(params), ptr(dth[0]), ptr(dth[1]), ptr(grads), ptr(ws[0]),
                          ptr(ws[1]), nb, stage, st, 0, device=cuda)
        else:
            for t in (0, 1):
                _lib.call("stabnet_tower_fwd_train", plan.handle, ptr(params), ptr(xs[t]), ptr(th[t]), ptr(ws[t]), nb, cfg.bn_eps,
                          cfg.bn_decay, st, 0, device=cuda)
            for t in (0, 1):
                _lib.call("stabnet_tower_bwd", plan.handle, ptr(params), ptr(dth[t]), ptr(grads), ptr(ws[t]), nb, st, 0, device=cuda)
        torch.cuda.synchronize()
        so, sh, mo, io = ctypes.c_long(), ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
        _lib.call("stabnet_net_train_bn_offsets", plan.handle, ctypes.byref(so), ctypes.byref(sh), ctypes.byref(mo), ctypes.byref(io))
        G = L.stabnet_net_bn_channels(plan.handle)
        bn = [w.view(torch.float32)[mo.value:mo.value + G].cpu().numpy() for w in ws]        # batch means of every BN, per tower
        return [t.cpu().numpy() for t in th], params.cpu().numpy(), grads.cpu().numpy(), bn

    th_l, p_l, g_l, bn_l = run(True)
    th_s, p_s, g_s, bn_s = run(False)
    for t in (0, 1):
        assert np.abs(th_l[t] - th_s[t]).max() < 2e-6 * max(1.0, np.abs(th_s[t]).max()), t         # float32 summation order only
        assert np.abs(bn_l[t] - bn_s[t]).max() < 1e-5 * max(1.0, np.abs(bn_s[t]).max()), t
    # the moving averages received tower 1's update, then tower 2's, in both forms
    assert np.abs(p_l - p_s).max() < 1e-6 * max(1.0, np.abs(p_s).max())
    # gradients: the same maps in another float32 summation order (pair launches pick their own split-K, the pair's wgrad slabs
    # are reduced as [tower][split]).  The backward amplifies rounding-sized differences -- the BN backward right behind
    # reduce_mean subtracts a per-channel constant from an almost constant gradient, batch statistics at these sizes cover
    # 12..96 values per channel -- measured 5.3e-3 (whole vector, relative L2) at 2 x 64 x 96; tests/test_train_gpu.py and
    # tests/test_baseline_sizes_gpu.py hold both forms against float64.  A wrong or missing tower in one launch is O(1) of its tensor.
    whole = np.linalg.norm(g_l - g_s) / np.linalg.norm(g_s)
    assert whole < 2e-2, whole
    gn = np.linalg.norm(g_s)
    for name, off, kind, dims, aux in plan.table:
        n = int(np.prod([d for d in dims if d > 0]))
        if off + n > nt or np.linalg.norm(g_s[off:off + n]) < 1e-3 * gn:
            continue
        e = np.linalg.norm(g_l[off:off + n] - g_s[off:off + n]) / np.linalg.norm(g_s[off:off + n])
        assert e < 1e-1, (name, e)
    cos = float(np.dot(g_l, g_s) / (np.linalg.norm(g_l) * np.linalg.norm(g_s)))
    assert cos > 1 - 2e-4, cos


def test_backward_must_match_the_forward_that_filled_the_workspace(cuda):
    """The lockstep forward keeps both towers' FC activations in the pair's first workspace, the single-tower forward in its
    own: a backward of the other kind would read rows that were never written.  The library refuses it (host-side stamp per
    workspace) instead of returning a silently wrong gradient."""
    from stabnet_amd import _lib, synthetic
    from stabnet_amd._tensor import ptr, stream_ptr
    from stabnet_amd.config import Config
    from stabnet_amd.regressor import NetPlan
    N, H, W = 2, 64, 96
    cfg = Config(height=H, width=W, batch_size=N)
    plan = NetPlan(N, H, W, cfg, keep_activations=True)
    params = torch.from_numpy(plan.pack(synthetic.make_params(cfg, seed=0, theta_scale=0.3))).to(cuda)
    rng = np.random.default_rng(3)
    xs = [torch.from_numpy(rng.uniform(-0.5, 0.5, (N, H, W, cfg.in_ch)).astype(np.float32)).to(cuda) for _ in range(2)]
    dth = [torch.from_numpy(rng.standard_normal((N, cfg.n_theta)).astype(np.float32)).to(cuda) for _ in range(2)]
    nb = _lib.lib().stabnet_net_train_workspace_bytes(plan.handle)
    ws = [torch.zeros(nb, dtype=torch.uint8, device=cuda) for _ in range(2)]
    th = [torch.empty((N, cfg.n_theta), dtype=torch.float32, device=cuda) for _ in range(2)]
    grads = torch.zeros(plan.n_trainable, dtype=torch.float32, device=cuda)
    st = stream_ptr(cuda)
    with pytest.raises(_lib.StabnetError, match="no training forward"):
        _lib.call("stabnet_tower_bwd", plan.handle, ptr(params), ptr(dth[0]), ptr(grads), ptr(ws[0]), nb, st, 0, device=cuda)
    _lib.call("stabnet_towers_fwd_train", plan.handle, ptr(params), ptr(xs[0]), ptr(xs[1]), ptr(th[0]), ptr(th[1]), ptr(ws[0]), ptr(ws[1]),
              nb, cfg.bn_eps, cfg.bn_decay, st, 0, device=cuda)
    with pytest.raises(_lib.StabnetError, match="stabnet_towers_bwd_stage"):
        _lib.call("stabnet_tower_bwd", plan.handle, ptr(params), ptr(dth[0]), ptr(grads), ptr(ws[0]), nb, st, 0, device=cuda)
    with pytest.raises(_lib.StabnetError, match="not filled together"):           # the pair in the wrong order
        _lib.call("stabnet_towers_bwd_stage", plan.handle, ptr(params), ptr(dth[0]), ptr(dth[1]), ptr(grads), ptr(ws[1]), ptr(ws[0]), nb, 0,
                  st, 0, device=cuda)
    _lib.call("stabnet_towers_bwd_stage", plan.handle, ptr(params), ptr(dth[0]), ptr(dth[1]), ptr(grads), ptr(ws[0]), ptr(ws[1]), nb, 0, st, 0,
              device=cuda)
    for t in (0, 1):
        _lib.call("stabnet_tower_fwd_train", plan.handle, ptr(params), ptr(xs[t]), ptr(th[t]), ptr(ws[t]), nb, cfg.bn_eps, cfg.bn_decay, st, 0,
                  device=cuda)
    with pytest.raises(_lib.StabnetError, match="not filled together"):
        _lib.call("stabnet_towers_bwd_stage", plan.handle, ptr(params), ptr(dth[0]), ptr(dth[1]), ptr(grads), ptr(ws[0]), ptr(ws[1]), nb, 0,
                  st, 0, device=cuda)
    _lib.call("stabnet_tower_bwd", plan.handle, ptr(params), ptr(dth[0]), ptr(grads), ptr(ws[0]), nb, st, 0, device=cuda)
    torch.cuda.synchronize()
