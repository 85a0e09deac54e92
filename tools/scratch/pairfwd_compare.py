import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from stabnet_amd import synthetic
from stabnet_amd.config import Config
from stabnet_amd.train import Trainer
N, H, W, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
cfg = Config(height=H, width=W, batch_size=N, max_matches=48) if os.environ.get('TESTCFG') else Config(height=H, width=W, batch_size=N)
P = synthetic.make_params(cfg, seed=0, theta_scale=0.3 if os.environ.get('TESTCFG') else 0.2)
tr = Trainer(P, N, H, W, cfg, device="cuda:0")
b = synthetic.make_train_batch(cfg, N, H, W, 5 if os.environ.get('TESTCFG') else 1234)
dev_b = {k: torch.from_numpy(v).to("cuda:0") for k, v in b.items()}
gates = {"use_theta_loss": 1, "use_temp_loss": 1, "use_black_loss": 1, "use_theta_only": 0}
if os.environ.get("PERTURB"):      # how sensitive are the gradients to a rounding-sized change of the input?
    torch.manual_seed(7)
    for k in ("x1", "x2"):
        dev_b[k] = dev_b[k] * (1.0 + float(os.environ["PERTURB"]) * torch.randn_like(dev_b[k]))
if os.environ.get("REGRESSOR_ONLY"):
    # fixed d_theta: no loss in the loop (the losses have discrete pixel choices: round(), strict black comparisons)
    torch.manual_seed(1)
    d1 = torch.randn(N, cfg.n_theta, device="cuda:0"); d2 = torch.randn(N, cfg.n_theta, device="cuda:0")
    tr.grads.zero_()
    th = tr._towers_fwd(dev_b["x1"], dev_b["x2"])
    tr._towers_bwd(d1, d2)
    torch.cuda.synchronize()
    print("theta", float(th[0].double().abs().sum()), float(th[1].double().abs().sum()))
else:
    tr.forward_backward(dev_b, gates, apply_update=False)
torch.cuda.synchronize()
if not os.environ.get("REGRESSOR_ONLY"):
    lo = tr.losses()
    print("BLACK", [float(t["black_pix"].double().sum()) for t in tr.last["towers"]])
    print("LOSSES", {k: (repr(v) if not isinstance(v, dict) else {kk: repr(vv) for kk, vv in v.items()}) for k, v in lo.items()})
g = tr.grad_flat().cpu().numpy()
np.save(out, g)
if len(sys.argv) > 5:
    ref = np.load(sys.argv[5])
    worst = []
    for name, off, kind, dims, aux in tr.plan.table:
        n = int(np.prod([d for d in dims if d > 0])) if off < tr.nt else 0
        if n == 0 or off + n > tr.nt or name.endswith('biases'): continue
        a, r = g[off:off + n], ref[off:off + n]
        sc = np.abs(r).max() + 1e-30
        worst.append((float(np.abs(a - r).max() / sc), name))
    if os.environ.get("IN_ORDER"):
        for name, off, kind, dims, aux in tr.plan.table:
            n = int(np.prod([d for d in dims if d > 0])) if off < tr.nt else 0
            if n == 0 or off + n > tr.nt or not name.endswith('weights'): continue
            a, r = g[off:off + n].astype(np.float64), ref[off:off + n].astype(np.float64)
            print("%.2e  %s" % (np.linalg.norm(a - r) / (np.linalg.norm(r) + 1e-300), name))
    print('WHOLE rel L2 diff %.3e' % (np.linalg.norm(g.astype(np.float64) - ref) / np.linalg.norm(ref.astype(np.float64))))
    worst.sort(reverse=True)
    for e, nme in worst[:16]: print("%.3e %s" % (e, nme))
    print('median', worst[len(worst)//2])
