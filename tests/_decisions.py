"""Test helper (not a test module): read back the discrete decisions a Trainer step really took -- ReLU signs of every
batch-norm site and FC layer, the max-pool arg-max, `black_pix`, the sampler's floor corners -- in the format
oracle/torch_ref.train_objective(decisions=...) takes, and compare them with the float64 evaluation's own."""
import ctypes

import numpy as np
import torch

from oracle import torch_ref as T


def _off(plan, what):
    from stabnet_amd import _lib
    off, cnt = ctypes.c_long(), ctypes.c_long()
    _lib.call("stabnet_net_train_debug_offset", plan.handle, what.encode(), ctypes.byref(off), ctypes.byref(cnt))
    return off.value, cnt.value


def gpu_decisions(tr):
    """tr: stabnet_amd.train.Trainer after forward_backward() -> {'1': {...}, '2': {...}}."""
    from stabnet_amd import _lib
    plan, N, H, W = tr.plan, tr.N, tr.H, tr.W
    so, sh, mo, io = ctypes.c_long(), ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
    _lib.call("stabnet_net_train_bn_offsets", plan.handle, ctypes.byref(so), ctypes.byref(sh), ctypes.byref(mo), ctypes.byref(io))
    gammas = [(name, off, dims[0]) for name, off, kind, dims, aux in plan.table if kind == 2]
    off_gamma = min(off for _, off, _ in gammas)
    fcx = [_off(plan, "fcx%d" % k) for k in range(4)]
    ws0 = tr.ws[0].view(torch.float32)
    out = {}
    for k, key in enumerate(("1", "2")):
        ws = tr.ws[k].view(torch.float32)
        relu = {}
        for name, off, C in gammas:
            chan = off - off_gamma
            toff, cnt = _off(plan, "bn:%d" % chan)
            x = ws[toff:toff + cnt].view(-1, C).double()
            scale, shift = ws[so.value + chan:so.value + chan + C].double(), ws[sh.value + chan:sh.value + chan + C].double()
            # the kernels decide on fma(x, scale, shift) > 0: the sign of the exactly rounded value = the sign of the exact one
            relu[name[:-len("/gamma")]] = (x * scale + shift > 0).cpu().numpy()
        for j in (1, 2, 3):                     # fcx[j] = relu(fc_j output) of the pair, tower rows [k*N, (k+1)*N)
            o, cnt = fcx[j]
            d = cnt // (2 * N)
            relu["fc%d" % j] = (ws0[o:o + cnt].view(2 * N, d)[k * N:(k + 1) * N] > 0).cpu().numpy()
        ao, acnt = _off(plan, "argmax")
        po, pcnt = _off(plan, "pool")
        am = tr.ws[k][4 * ao:4 * ao + acnt].cpu().numpy()
        tw = tr.last["towers"][k]
        xm, ym = tw["x_map"].reshape(N, H, W).cpu().numpy(), tw["y_map"].reshape(N, H, W).cpu().numpy()
        out[key] = {"relu": relu, "pool_argmax": am.reshape(N, -1), "black": tw["black_pix"].reshape(N, H, W).cpu().numpy() != 0,
                    "corners": T.corners_f32(xm, ym, H, W)}
    return out


def flips(gpu, own):
    """[(tower, site, count, total)] of the decisions that differ between the forward under test and the float64 evaluation."""
    res = []
    for key in ("1", "2"):
        g, o = gpu[key], own[key]
        for site in sorted(g["relu"]):
            a, b = np.asarray(g["relu"][site]).ravel() != 0, np.asarray(o["relu"][site]).ravel() != 0
            if (a != b).any():
                res.append((key, "relu " + site, int((a != b).sum()), a.size))
        a, b = np.asarray(g["pool_argmax"]).ravel(), np.asarray(o["pool_argmax"]).ravel()
        if (a != b).any():
            res.append((key, "pool_argmax", int((a != b).sum()), a.size))
        a, b = np.asarray(g["black"]).ravel(), np.asarray(o["black"]).ravel()
        if (a != b).any():
            res.append((key, "black_pix", int((a != b).sum()), a.size))
        n = sum(int((np.asarray(x).ravel() != np.asarray(y).ravel()).sum()) for x, y in zip(g["corners"], o["corners"]))
        if n:
            res.append((key, "sampler corners", n, 4 * np.asarray(g["corners"][0]).size))
    return res


def gradient_errors(plan, got_flat, want_flat):
    """(worst element error relative to its tensor's gradient scale, worst tensor relative L2, whole-gradient relative L2, per-tensor rows)."""
    gmax = np.abs(want_flat).max()
    worst_abs = worst_l2 = 0.0
    rows = []
    for name, off, kind, dims, aux in plan.table:
        if kind in (4, 5):
            continue
        n = int(np.prod([d for d in dims if d > 0]))
        gg, ww = got_flat[off:off + n].astype(np.float64), want_flat[off:off + n].astype(np.float64)
        # tensors whose gradient is analytically ~0 (e.g. a bias in front of a batch-stat BN) are judged against the
        # global gradient scale instead of their own
        scale = max(np.abs(ww).max(), 1e-5 * gmax)
        err = np.abs(gg - ww).max() / scale
        l2 = np.linalg.norm(gg - ww) / max(np.linalg.norm(ww), 1e-5 * gmax * np.sqrt(n))
        rows.append((name, err, l2))
        worst_abs, worst_l2 = max(worst_abs, err), max(worst_l2, l2)
    return worst_abs, worst_l2, float(np.linalg.norm(got_flat - want_flat) / np.linalg.norm(want_flat)), rows
