import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from stabnet_amd import synthetic, _lib
from stabnet_amd.config import Config
from stabnet_amd.deploy import StabNetStream
g = np.load(os.path.join(ROOT, "tests/golden/clip_256x256_t64.npz"))
H, W, Tn, clip_seed, weight_seed, stride = (int(v) for v in g["meta"])
cfg = Config(height=H, width=W)
P = synthetic.make_params(cfg, seed=weight_seed, theta_scale=float(g["theta_scale"]))
clip = torch.from_numpy(synthetic.make_clip(H, W, Tn, seed=clip_seed, margin=64)).cuda()
if os.environ.get("NOTABLE"):
    _lib.lib().stabnet_conv_tuning_table_set(-1, 0, 0, 0, 0, 0)
s = StabNetStream(P, H, W, cfg, streams=1, use_graph=False)
s.start(clip[0:1])
errs = []
for t in range(1, Tn):
    r = s.step(clip[t:t + 1])
    errs.append(float(np.abs(r["theta"].cpu().numpy()[0] - g["theta"][t - 1]).max()))
print("theta err per frame:", " ".join("%.1e" % e for e in errs))
