"""Child of tests/test_pair_forward_gpu.py (not a test module): one lockstep training forward + backward of both towers with
the environment's conv switches; dumps every kept activation of both towers, the batch-norm statistics, theta and the
gradient to <out>.npz."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out, N, H, W):
    from stabnet_amd import _lib, synthetic
    from stabnet_amd._tensor import ptr, stream_ptr
    from stabnet_amd.config import Config
    from stabnet_amd.regressor import NetPlan
    dev = torch.device("cuda", 0)
    cfg = Config(height=H, width=W, batch_size=N)
    plan = NetPlan(N, H, W, cfg, keep_activations=True)
    flat = plan.pack(synthetic.make_params(cfg, seed=0, theta_scale=0.3))
    rng = np.random.default_rng(3)
    xs = [torch.from_numpy(rng.uniform(-0.5, 0.5, (N, H, W, cfg.in_ch)).astype(np.float32)).to(dev) for _ in range(2)]
    dth = [torch.from_numpy(rng.standard_normal((N, cfg.n_theta)).astype(np.float32)).to(dev) for _ in range(2)]
    L = _lib.lib()
    nb = L.stabnet_net_train_workspace_bytes(plan.handle)
    st = stream_ptr(dev)
    params = torch.from_numpy(flat.copy()).to(dev)
    ws = [torch.zeros(nb, dtype=torch.uint8, device=dev) for _ in range(2)]
    th = [torch.empty((N, cfg.n_theta), dtype=torch.float32, device=dev) for _ in range(2)]
    grads = torch.zeros(plan.n_trainable, dtype=torch.float32, device=dev)
    _lib.call("stabnet_towers_fwd_train", plan.handle, ptr(params), ptr(xs[0]), ptr(xs[1]), ptr(th[0]), ptr(th[1]), ptr(ws[0]), ptr(ws[1]),
              nb, cfg.bn_eps, cfg.bn_decay, st, 0, device=dev)
    torch.cuda.synchronize()
    so, sh, mo, io = ctypes.c_long(), ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
    _lib.call("stabnet_net_train_bn_offsets", plan.handle, ctypes.byref(so), ctypes.byref(sh), ctypes.byref(mo), ctypes.byref(io))
    G = L.stabnet_net_bn_channels(plan.handle)
    nact = so.value                      # the activation region ends where the batch-norm buffers begin
    acts = np.stack([w.view(torch.float32)[:nact].cpu().numpy() for w in ws])
    bn = np.stack([np.concatenate([w.view(torch.float32)[o.value:o.value + G].cpu().numpy() for o in (so, sh, mo, io)]) for w in ws])
    for stage in range(L.stabnet_net_num_grad_stages()):
        _lib.call("stabnet_towers_bwd_stage", plan.handle, ptr(params), ptr(dth[0]), ptr(dth[1]), ptr(grads), ptr(ws[0]), ptr(ws[1]), nb,
                  stage, st, 0, device=dev)
    torch.cuda.synchronize()
    np.savez(out, acts=acts.view(np.uint32), bn=bn.view(np.uint32), theta=np.stack([t.cpu().numpy() for t in th]).view(np.uint32),
             grads=grads.cpu().numpy().view(np.uint32), params=params.cpu().numpy().view(np.uint32))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
