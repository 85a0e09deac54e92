"""Dump the training forward's workspaces (activation region) of both towers; with a reference dump: report where they differ."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
sys.path.insert(0, os.path.join(ROOT, "tests"))
from stabnet_amd import synthetic, _lib
from stabnet_amd.config import Config
from stabnet_amd.train import Trainer
N, H, W, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
cfg = Config(height=H, width=W, batch_size=N, max_matches=48)
P = synthetic.make_params(cfg, seed=0, theta_scale=0.3)
b = synthetic.make_train_batch(cfg, N, H, W, 5)
tr = Trainer(P, N, H, W, cfg, device="cuda:0")
x1 = torch.from_numpy(b["x1"]).cuda(); x2 = torch.from_numpy(b["x2"]).cuda()
for w in tr.ws: w.zero_()
th = tr._towers_fwd(x1, x2)
torch.cuda.synchronize()
nact = _lib.lib().stabnet_net_workspace_bytes(tr.plan.handle) // 4
acts = [w.view(torch.float32)[:nact].cpu().numpy() for w in tr.ws]
np.save(out, np.stack(acts))
print("theta", [float(t.double().abs().sum()) for t in th])
if len(sys.argv) > 5:
    ref = np.load(sys.argv[5])
    for t in (0, 1):
        d = np.abs(acts[t] - ref[t])
        bad = np.nonzero(d > 1e-4 * (np.abs(ref[t]) + 1e-3))[0]
        print("tower", t, "max abs diff %.3e" % d.max(), "count > 1e-4 rel:", bad.size, "first offsets", bad[:5], "last", bad[-3:] if bad.size else None)
        # histogram of differing offsets in 16 equal bins of the region
        if bad.size:
            h, edges = np.histogram(bad, bins=16, range=(0, nact))
            print("   bins", h.tolist())
