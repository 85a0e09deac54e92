"""Child of tests/test_b2b_plan_gpu.py (not a test module): the inference regressor of one synthetic input with the plan
switches of the environment (read once per process); dumps theta and the number of launches to <out>.npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out, N, H, W):
    from stabnet_amd import _lib, synthetic
    from stabnet_amd.config import Config
    from stabnet_amd.regressor import Regressor
    cfg = Config(height=H, width=W)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.3)
    rng = np.random.default_rng(11)
    x = rng.uniform(-0.5, 0.5, (N, H, W, cfg.in_ch)).astype(np.float32)
    reg = Regressor(P, N, H, W, cfg, device="cuda:0")
    theta = reg(torch.from_numpy(x).cuda())
    theta2 = reg(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    np.savez(out, theta=theta.cpu().numpy(), theta2=theta2.cpu().numpy(), x=x,
             launches=np.int64(_lib.lib().stabnet_net_num_launches(reg.plan.handle)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
