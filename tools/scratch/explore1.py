import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import stabnet_oracle as O
from oracle import torch_ref as T
from stabnet_amd import synthetic
from stabnet_amd.config import Config
from stabnet_amd.train import Trainer
cuda = torch.device("cuda:0")
# (A) loss trajectories
N, H, W = 2, 64, 96
for lr in (2e-5, 5e-5, 1e-4):
    cfg = Config(height=H, width=W, batch_size=N, max_matches=48, initial_learning_rate=lr)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.3)
    b = synthetic.make_train_batch(cfg, N, H, W, 5)
    dev_b = {k: torch.from_numpy(v).to(cuda) for k, v in b.items()}
    gates = {"use_theta_loss": 1, "use_temp_loss": 1, "use_black_loss": 1, "use_theta_only": 0}
    tr = Trainer(P, N, H, W, cfg, device=cuda)
    losses = []
    for _ in range(40):
        tr.forward_backward(dev_b, gates)
        losses.append(round(tr.losses()["total_loss"], 3))
    print("lr", lr, losses, flush=True)
# (B) gradient errors at 8x288x512
N, H, W = int(os.environ.get("NB", "8")), 288, 512
cfg = Config(height=H, width=W, batch_size=N, max_matches=512)
ocfg = O.Config(height=H, width=W, batch_size=N, max_matches=512)
P = synthetic.make_params(cfg, seed=0, theta_scale=0.3)
b = synthetic.make_train_batch(cfg, N, H, W, 1234)
b["flow"] = (b["flow"] + np.random.default_rng(1).normal(0, 0.01, b["flow"].shape)).astype(np.float32)
gates = {"use_theta_loss": 1, "use_temp_loss": 1, "use_black_loss": 1, "use_theta_only": 0}
tr = Trainer(P, N, H, W, cfg, device=cuda)
dev_b = {k: torch.from_numpy(v).to(cuda) for k, v in b.items()}
tr.forward_backward(dev_b, gates, apply_update=False)
torch.cuda.synchronize()
got_flat = tr.grad_flat().cpu().numpy()
pt = {k: T.t(v, requires_grad=True) for k, v in P.items()}
tot64, parts = T.train_objective(pt, b, ocfg, 1.0, 1.0, 0.0, training=True)
tot64.backward()
want_flat = tr.plan.pack({k: (pt[k].grad.numpy() if pt[k].grad is not None else np.zeros(P[k].shape)) for k in P})[:tr.nt]
gmax = np.abs(want_flat).max()
rows = []
for name, off, kind, dims, aux in tr.plan.table:
    if kind in (4, 5):
        continue
    n = int(np.prod([d for d in dims if d > 0]))
    gg, ww = got_flat[off:off + n].astype(np.float64), want_flat[off:off + n].astype(np.float64)
    scale = max(np.abs(ww).max(), 1e-5 * gmax)
    rows.append((np.abs(gg - ww).max() / scale, np.linalg.norm(gg - ww) / max(np.linalg.norm(ww), 1e-30), np.abs(ww).max(), name))
rows.sort(reverse=True)
for r in rows[:25]:
    print("%.4f  l2rel %.5f  scale %.3g  %s" % r)
print("theta err", [float(np.abs(tr.theta[k].cpu().numpy() - parts["tower%d" % (k + 1)]["theta"].detach().numpy()).max()) for k in (0, 1)])
cos = float(np.dot(got_flat, want_flat) / (np.linalg.norm(got_flat) * np.linalg.norm(want_flat)))
print("cos", cos, "gmax", gmax, "l2rel all", np.linalg.norm(got_flat - want_flat) / np.linalg.norm(want_flat))
