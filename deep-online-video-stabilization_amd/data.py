"""Training sample assembly on the device -- host mirror of get_data_mini_after.py's tensor pipeline.

The reference builds each sample inside TF input queues on the CPU (read_and_decode, get_data_mini_after.py:158-253) and
draws its random numbers from TF's Philox streams.  Here the draws are made by the caller's `numpy.random.Generator`
(get_rand_para / get_rand_H below keep the reference's ranges and its shared-seed behaviour) and the arithmetic runs in
libstabnet_hip.so (csrc/augment.hip) for a whole batch of pairs at once."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._tensor import dev_f32, ptr, stream_ptr
from .config import Config


def resized_hw(cfg: Config, H: int, W: int):
    return int(H / cfg.random_crop_rate), int(W / cfg.random_crop_rate)      # get_data_mini_after.py:8-9


def get_rand_para(x0: int, cfg: Config, H: int, W: int):
    """get_rand_para (get_data_mini_after.py:7-12): crop offsets and the flip flag derived from them.  read_and_decode hands
    the SAME python `seed` to every random op of a sample (:228-238), and TF1 gives ops with equal (graph, op) seeds the same
    Philox stream, so all of them consume the same first 32-bit output `x0`.  [external] RandomUniformInt: lo + x0 % range."""
    h, w = resized_hw(cfg, H, W)
    hh = int(x0 % (h - H)) if h > H else 0
    ww = int(x0 % (w - W)) if w > W else 0
    return {"h": hh, "w": ww, "flip": (hh + ww) % 2}


def get_jitter(x0: int):
    """tf.image.random_contrast(0.5, 1.5, seed) and random_brightness(32/255, seed) (get_data_mini_after.py:24-25), every
    channel of the pair alike (same seed again).  [external] RandomUniform float32: u = bitcast(0x3f800000 | (x0 & 0x7fffff)) - 1,
    value = lo + u*(hi - lo)."""
    u = np.float32((x0 & 0x7FFFFF) / float(1 << 23))
    contrast = np.float32(0.5) + u * np.float32(1.0)
    md = np.float32(32.0 / 255.0)
    bright = -md + u * (md - (-md))
    return np.float32(contrast), np.float32(bright)


def get_rand_H(rng: np.random.Generator, cfg: Config, n: int):
    """get_rand_H (get_data_mini_after.py:72-82) for n channels: independent uniforms per entry; with rand_H_change_rate = 1
    every channel gets a fresh matrix (H*rate + last_H*(1-rate))."""
    lo, hi = np.asarray(cfg.rand_H_min, np.float32), np.asarray(cfg.rand_H_max, np.float32)
    out = np.empty((n, 3, 3), np.float32)
    last = np.zeros((3, 3), np.float32)
    for i in range(n):
        Hm = rng.uniform(lo, hi).astype(np.float32)
        if i > 0:
            Hm = (Hm * np.float32(cfg.rand_H_change_rate) + last * np.float32(1 - cfg.rand_H_change_rate)).astype(np.float32)
        out[i] = last = Hm
    return out


def draw(rng: np.random.Generator, cfg: Config, N: int, H: int, W: int):
    """All random inputs of `augment_pairs` for N pairs, as host arrays."""
    para = np.empty((N, 3), np.int32)
    jitter = np.empty((N, 2), np.float32)
    Hs = np.empty((N, 2, cfg.before_ch, 9), np.float32)
    for n in range(N):
        x0 = int(rng.integers(0, 1 << 32, dtype=np.uint64))     # stands in for the sample's first Philox output
        p = get_rand_para(x0, cfg, H, W)
        para[n] = (p["h"], p["w"], p["flip"])
        jitter[n] = get_jitter(x0)
        Hs[n, 0] = get_rand_H(rng, cfg, cfg.before_ch).reshape(cfg.before_ch, 9)
        Hs[n, 1] = get_rand_H(rng, cfg, cfg.before_ch).reshape(cfg.before_ch, 9)
    return para, jitter, Hs


def augment_pairs(stable, unstable, flow, matches1, n1, matches2, n2, para, jitter, Hs, cfg: Config):
    """read_and_decode's output tuple (x1, y1, x2, y2, flow, feature_matches1, mask1, feature_matches2, mask2) for a batch,
    every tensor on the device.  stable [N,H,W,2*(before_ch+1)], unstable [N,H,W,2], flow [N,H,W,2], matches [N,M,4],
    n [N] int32 valid counts; para/jitter/Hs as returned by `draw` (host arrays or device tensors)."""
    stable = dev_f32(stable, "stable")
    dev = stable.device
    N, H, W, C = stable.shape
    bc = cfg.before_ch
    assert C == 2 * (bc + 1) and cfg.input_mask, "stable must hold 2*(before_ch+1) channels; input_mask configs only"
    t = lambda a, dt: a.to(dev) if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    unstable = dev_f32(unstable, "unstable")
    para_d, jit_d, Hs_d = t(para, np.int32), t(jitter, np.float32), t(Hs, np.float32)
    flow_d = dev_f32(flow, "flow") if flow is not None else None
    M = cfg.max_matches
    x1 = torch.empty((N, H, W, 2 * bc + 1), dtype=torch.float32, device=dev); x2 = torch.empty_like(x1)
    y1 = torch.empty((N, H, W, 1), dtype=torch.float32, device=dev); y2 = torch.empty_like(y1)
    flow_o = torch.empty((N, H, W, 2), dtype=torch.float32, device=dev) if flow is not None else None
    have_m = matches1 is not None
    if have_m:
        m1, m2 = dev_f32(matches1, "matches1"), dev_f32(matches2, "matches2")
        assert m1.shape == (N, M, 4) and m2.shape == (N, M, 4)
        c1, c2 = t(n1, np.int32), t(n2, np.int32)
        fm1 = torch.empty_like(m1); fm2 = torch.empty_like(m2)
        mk1 = torch.empty((N, M), dtype=torch.float32, device=dev); mk2 = torch.empty_like(mk1)
    else:
        m1 = m2 = c1 = c2 = fm1 = fm2 = mk1 = mk2 = None
    L = _lib.lib()
    nbytes = L.stabnet_augment_workspace_bytes(N, H, W, bc)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.call("stabnet_augment_pairs", ptr(stable), ptr(unstable), ptr(flow_d), ptr(m1), ptr(c1), ptr(m2), ptr(c2), ptr(para_d),
              ptr(jit_d), ptr(Hs_d), N, H, W, bc, M, float(cfg.random_crop_rate), ptr(x1), ptr(y1), ptr(x2), ptr(y2), ptr(flow_o),
              ptr(fm1), ptr(mk1), ptr(fm2), ptr(mk2), ptr(ws), nbytes, stream_ptr(dev), device=dev)
    return x1, y1, x2, y2, flow_o, fm1, mk1, fm2, mk2
