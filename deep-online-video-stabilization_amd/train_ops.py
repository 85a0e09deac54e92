"""Host wrappers of the training-side C-ABI entry points (backward of warp/sampler, loss kernels)."""
from __future__ import annotations

import torch

from . import _lib
from ._tensor import dev_f32, empty, ptr, stream_ptr
from .config import Config, v2_93


def transformer_bwd(pts2, Hs, U, x_map, y_map, d_out=None, d_xmap=None, d_ymap=None, cfg: Config = v2_93, dmap_scale=None):
    """d_xmap / d_ymap are multiplied per sample by dmap_scale [N] when given (feature_loss returns counts + that scale)."""
    U = dev_f32(U, "U")
    N, H, W, C = U.shape
    d_pts2 = empty((N, cfg.grid_h + 1, cfg.grid_w + 1, 2), U)
    ws = empty((N * cfg.grid_h * cfg.grid_w * 8 + 1,), U, dtype=torch.int64)    # fixed-point sums + the poison word
    _lib.call("stabnet_transformer_bwd", ptr(dev_f32(pts2)), ptr(dev_f32(Hs)), ptr(U), ptr(dev_f32(x_map)),
              ptr(dev_f32(y_map)), ptr(d_out), ptr(d_xmap), ptr(d_ymap), ptr(dmap_scale), N, H, W, C, cfg.grid_h, cfg.grid_w, ptr(d_pts2),
              ptr(ws), stream_ptr(U.device), device=U.device)
    return d_pts2


def interp_bwd(x, y, d_out, d_im=None):
    """d_im (+)= scatter of d_out; a given d_im is accumulated into."""
    d_out = dev_f32(d_out, "d_out")
    N, H, W, C = d_out.shape
    acc = d_im is not None
    if d_im is None:
        d_im = empty((N, H, W, C), d_out)
    ws = empty((N * H * W * C + 1,), d_out, dtype=torch.int64)       # fixed-point accumulators of the scatter + the poison word
    _lib.call("stabnet_interp_bwd", ptr(dev_f32(x)), ptr(dev_f32(y)), ptr(d_out), N, H, W, C, ptr(d_im), int(acc), ptr(ws),
              stream_ptr(d_out.device), device=d_out.device)
    return d_im


def axpb(x, a, b):
    x = dev_f32(x)
    y = torch.empty_like(x)
    _lib.call("stabnet_axpb", ptr(x), float(a), float(b), x.numel(), ptr(y), stream_ptr(x.device), device=x.device)
    return y


def masked_mse_sums(a, b, black, m2=None):
    a = dev_f32(a)
    N = a.shape[0]
    hw = a.numel() // N
    sums = empty((N, 2), a)
    ws = torch.empty(_lib.lib().stabnet_masked_mse_workspace_bytes(N), dtype=torch.uint8, device=a.device)
    _lib.call("stabnet_masked_mse_sums", ptr(a), ptr(dev_f32(b)), ptr(dev_f32(black)), ptr(m2), N, hw, ptr(sums), ptr(ws),
              stream_ptr(a.device), device=a.device)
    return sums


def masked_mse_grad(a, b, black, m2, sums, coef, ga=None, accumulate_a=False, want_gb=False):
    a = dev_f32(a)
    N = a.shape[0]
    hw = a.numel() // N
    if ga is None:
        ga = torch.empty_like(a)
        accumulate_a = False
    gb = torch.empty_like(a) if want_gb else None
    _lib.call("stabnet_masked_mse_grad", ptr(a), ptr(dev_f32(b)), ptr(dev_f32(black)), ptr(m2), ptr(sums), float(coef),
              N, hw, ptr(ga), int(accumulate_a), ptr(gb), stream_ptr(a.device), device=a.device)
    return ga, gb


def feature_loss(matches, mask, x_map, y_map, gcoef=0.0, want_grad=False, want_warped=False):
    """-> (value [N], d_xmap, d_ymap, warped, dscale): the map gradient is (d_xmap, d_ymap) [signed counts] * dscale[n]."""
    matches = dev_f32(matches)
    N, Mx, _ = matches.shape
    x_map = dev_f32(x_map)
    H, W = x_map.shape[1], x_map.shape[2]
    value = empty((N,), matches)
    dxm = empty((N, H, W), matches) if want_grad else None
    dym = empty((N, H, W), matches) if want_grad else None
    warped = empty((N, Mx, 2), matches) if want_warped else None
    dscale = empty((N,), matches) if want_grad else None
    _lib.call("stabnet_feature_loss", ptr(matches), ptr(dev_f32(mask)), ptr(x_map), ptr(dev_f32(y_map)), N, H, W, Mx,
              float(gcoef), ptr(value), ptr(dxm), ptr(dym), ptr(dscale), ptr(warped), stream_ptr(matches.device),
              device=matches.device)
    return value, dxm, dym, warped, dscale


def mesh_losses(theta, d_pts2_warp, cfg: Config, w_id, w_dist, w_cons, use_black, w_black):
    theta = dev_f32(theta)
    N = theta.shape[0]
    losses = empty((4,), theta)
    d_theta = torch.empty_like(theta)
    _lib.call("stabnet_mesh_losses", ptr(theta), ptr(d_pts2_warp), N, cfg.grid_h, cfg.grid_w, cfg.do_crop_rate,
              cfg.id_mul, float(w_id), float(w_dist), float(w_cons), float(use_black), float(w_black), ptr(losses),
              ptr(d_theta), stream_ptr(theta.device), device=theta.device)
    return losses, d_theta
