import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from stabnet_amd import synthetic
from stabnet_amd.config import Config
from stabnet_amd.train import Trainer
N, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = Config(height=H, width=W, batch_size=N)
P = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
tr = Trainer(P, N, H, W, cfg, device="cuda:0")
print("trainer built", flush=True)
torch.manual_seed(0)
x1 = torch.randn(N, H, W, 13, device="cuda:0"); x2 = torch.randn(N, H, W, 13, device="cuda:0")
th = tr._towers_fwd(x1, x2)
torch.cuda.synchronize()
print("forward ok", [float(t.abs().mean()) for t in th], flush=True)
