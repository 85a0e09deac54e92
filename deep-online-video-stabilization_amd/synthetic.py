"""Seeded synthetic weights and inputs.  There is no checkpoint, .meta graph or
dataset offline, so every run uses these.  NumPy only; arrays are in the reference's (TF) layouts:
conv weights HWIO, FC weights [in,out], names as under `stable_net/resnet/`."""
from __future__ import annotations

import numpy as np

from .config import Config

RESNET_V2_50_BLOCKS = (
    ("block1", 256, 64, 3, 2),
    ("block2", 512, 128, 4, 2),
    ("block3", 1024, 256, 6, 2),
    ("block4", 2048, 512, 3, 1),
)


def param_spec(cfg: Config):
    """Ordered list of (name, shape, kind); kind in conv_w, bias, gamma, beta, mean, var, fc_w, fc_b."""
    spec = []
    R = "resnet_v2_50/"

    def bn(prefix, c):
        spec.extend([(prefix + "/gamma", (c,), "gamma"), (prefix + "/beta", (c,), "beta"),
                     (prefix + "/moving_mean", (c,), "mean"), (prefix + "/moving_variance", (c,), "var")])

    spec.append((R + "conv1/weights", (7, 7, cfg.in_ch, 64), "conv_w"))
    spec.append((R + "conv1/biases", (64,), "bias"))
    cin = 64
    for (bname, depth, dbn, units, _s) in RESNET_V2_50_BLOCKS:
        for u in range(1, units + 1):
            S = R + "%s/unit_%d/bottleneck_v2/" % (bname, u)
            bn(S + "preact", cin)
            if cin != depth:
                spec.append((S + "shortcut/weights", (1, 1, cin, depth), "conv_w"))
                spec.append((S + "shortcut/biases", (depth,), "bias"))
            spec.append((S + "conv1/weights", (1, 1, cin, dbn), "conv_w"))
            bn(S + "conv1/BatchNorm", dbn)
            spec.append((S + "conv2/weights", (3, 3, dbn, dbn), "conv_w"))
            bn(S + "conv2/BatchNorm", dbn)
            spec.append((S + "conv3/weights", (1, 1, dbn, depth), "conv_w3"))
            spec.append((S + "conv3/biases", (depth,), "bias"))
            cin = depth
    bn(R + "postnorm", cin)
    dims = [cin, 2048, 1024, 512]
    for k in (1, 2, 3):
        spec.append(("fc/fc/fc_%d/weights" % k, (dims[k - 1], dims[k]), "fc_w"))
        spec.append(("fc/fc/fc_%d/biases" % k, (dims[k],), "fc_b"))
    spec.append(("fc/fc_weights", (512, cfg.n_theta), "out_w"))
    spec.append(("fc/fc_bias", (cfg.n_theta,), "out_b"))
    return spec


def make_params(cfg: Config, seed: int = 0, theta_scale: float = 0.02):
    """Seeded weights: fan-in normal convs (conv3 of each unit damped so the residual stream stays O(1..10)),
    non-trivial BN affine and moving statistics, Xavier-uniform FCs, output layer scaled so theta ~ O(0.05)."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape, kind in param_spec(cfg):
        if kind in ("conv_w", "conv_w3"):
            fan_in = shape[0] * shape[1] * shape[2]
            w = rng.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_in))
            if kind == "conv_w3":
                w *= np.float32(0.3)
            p[name] = w
        elif kind == "bias":
            p[name] = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.01)
        elif kind == "gamma":
            p[name] = rng.uniform(0.8, 1.2, shape).astype(np.float32)
        elif kind == "beta":
            p[name] = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
        elif kind == "mean":
            p[name] = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
        elif kind == "var":
            p[name] = rng.uniform(0.8, 1.2, shape).astype(np.float32)
        elif kind == "fc_w":
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            p[name] = rng.uniform(-lim, lim, shape).astype(np.float32)
        elif kind == "fc_b":
            p[name] = np.zeros(shape, np.float32)
        elif kind == "out_w":
            lim = np.sqrt(3.0 / shape[0])
            p[name] = (rng.uniform(-lim, lim, shape) * theta_scale).astype(np.float32)
        elif kind == "out_b":
            p[name] = np.zeros(shape, np.float32)
    return p


def _box_blur(a, k=9):
    c = np.cumsum(np.pad(a, ((k // 2 + 1, k // 2), (0, 0)), mode="edge"), axis=0)
    a = (c[k:] - c[:-k]) / k
    c = np.cumsum(np.pad(a, ((0, 0), (k // 2 + 1, k // 2)), mode="edge"), axis=1)
    return (c[:, k:] - c[:, :-k]) / k


def make_clip(H: int, W: int, T: int, seed: int = 1234, margin: int = 64):
    """Shaky synthetic clip: jittered crops of a blurred-noise texture in [-0.5, 0.5]. -> [T,H,W] float32."""
    rng = np.random.default_rng(seed)
    base = rng.random((H + 2 * margin, W + 2 * margin))
    base = _box_blur(_box_blur(base))
    base = (base - base.min()) / (base.max() - base.min()) - 0.5
    d = np.clip(np.rint(np.cumsum(rng.normal(0.0, 1.5, (T, 2)), axis=0)), -(margin - 16), margin - 16).astype(int)
    out = np.empty((T, H, W), np.float32)
    for t in range(T):
        dy, dx = d[t]
        out[t] = base[margin + dy:margin + dy + H, margin + dx:margin + dx + W]
    return out


def _rand_mask(rng, H, W, cfg: Config):
    """History mask of one lag: out-of-frame region of a random homography (get_data_mini_after.py:93-108 shape)."""
    hmax = np.array([[1.1, 0.1, 0.5], [0.1, 1.1, 0.5], [0.1, 0.1, 1]])
    hmin = np.array([[0.9, -0.1, -0.5], [-0.1, 0.9, -0.5], [-0.1, -0.1, 1]])
    Hm = hmin + (hmax - hmin) * rng.random((3, 3)) * 0.3 + (np.eye(3) - (hmin + (hmax - hmin) * 0.5)) * 0.7
    xs, ys = np.meshgrid(np.linspace(-1, 1, W), np.linspace(-1, 1, H))
    z = Hm[2, 0] * xs + Hm[2, 1] * ys + Hm[2, 2]
    x = (Hm[0, 0] * xs + Hm[0, 1] * ys + Hm[0, 2]) / z
    y = (Hm[1, 0] * xs + Hm[1, 1] * ys + Hm[1, 2]) / z
    return ((x < -1) | (x > 1) | (y < -1) | (y > 1)).astype(np.float32)


def make_stack(cfg: Config, N: int, H: int, W: int, seed: int = 1234):
    """One tower input [N,H,W,13]: 6 masks, 6 history frames (masked pixels = -1), current frame; plus label y."""
    rng = np.random.default_rng(seed)
    x = np.empty((N, H, W, cfg.in_ch), np.float32)
    y = np.empty((N, H, W, 1), np.float32)
    for n in range(N):
        clip = make_clip(H, W, cfg.before_ch + 2, seed=seed * 1000 + n, margin=32)
        for c in range(cfg.before_ch):
            m = _rand_mask(rng, H, W, cfg)
            x[n, :, :, c] = m
            x[n, :, :, cfg.before_ch + c] = clip[c] * (1 - m) - m
        x[n, :, :, 2 * cfg.before_ch] = clip[cfg.before_ch]
        y[n, :, :, 0] = clip[cfg.before_ch + 1]
    return x, y


def make_train_batch(cfg: Config, N: int, H: int, W: int, seed: int = 1234):
    """One siamese training batch: x1,y1,x2,y2,flow,matches1,mask1,matches2,mask2."""
    rng = np.random.default_rng(seed + 7)
    x1, y1 = make_stack(cfg, N, H, W, seed)
    x2, y2 = make_stack(cfg, N, H, W, seed + 1)
    gx, gy = np.meshgrid(np.linspace(-1, 1, W, dtype=np.float32), np.linspace(-1, 1, H, dtype=np.float32))
    flow = np.stack([gx, gy], axis=2)[None].repeat(N, 0) + rng.normal(0, 0.01, (N, H, W, 2)).astype(np.float32)
    out = {"x1": x1, "y1": y1, "x2": x2, "y2": y2, "flow": flow.astype(np.float32)}
    for k in ("1", "2"):
        out["matches" + k] = rng.uniform(-1, 1, (N, cfg.max_matches, 4)).astype(np.float32)
        m = np.zeros((N, cfg.max_matches), np.float32)
        m[:, :500] = 1
        out["mask" + k] = m
    return out


def make_raw_pairs(cfg: Config, N: int, H: int, W: int, seed: int = 1234):
    """Un-augmented pair material in the layout read_and_decode assembles before its random ops
    (get_data_mini_after.py:177-213): stable [N,H,W,2*(before_ch+1)] = (label, before_ch history frames) of tower 1 then
    tower 2, unstable [N,H,W,2] current frames, flow [N,H,W,2], matches [N,max_matches,4] zero-padded + valid counts.
    Feed to stabnet_amd.data.augment_pairs."""
    rng = np.random.default_rng(seed + 11)
    bc = cfg.before_ch
    stable = np.empty((N, H, W, 2 * (bc + 1)), np.float32)
    unstable = np.empty((N, H, W, 2), np.float32)
    for n in range(N):
        st = make_clip(H, W, bc + 2, seed=seed * 1000 + n, margin=32)            # smooth camera path
        sh = make_clip(H, W, 2, seed=seed * 1000 + n + 500, margin=32)            # the shaky views of the two current frames
        for tower in range(2):
            stable[n, :, :, tower * (bc + 1)] = st[bc + tower]                     # label = stable frame at the current time
            for k in range(bc):
                stable[n, :, :, tower * (bc + 1) + 1 + k] = st[max(bc - 1 - k + tower, 0)]
            unstable[n, :, :, tower] = sh[tower]
    gx, gy = np.meshgrid(np.linspace(-1, 1, W, dtype=np.float32), np.linspace(-1, 1, H, dtype=np.float32))
    flow = (np.stack([gx, gy], axis=2)[None].repeat(N, 0) + rng.normal(0, 0.01, (N, H, W, 2))).astype(np.float32)
    out = {"stable": stable, "unstable": unstable, "flow": flow}
    for k in ("1", "2"):
        cnt = rng.integers(400, 600, N).astype(np.int32)
        m = rng.uniform(-1, 1, (N, cfg.max_matches, 4)).astype(np.float32)
        for n in range(N):
            m[n, cnt[n]:] = 0
        out["matches" + k] = m
        out["n" + k] = cnt
    return out
