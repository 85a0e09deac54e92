"""Host mirror of the reference's warp operators over the C ABI (same names, argument meaning, returns).

  transformer(U, theta)            spatial_transformer3.py:19   -> (output, black_pix, img)
  interpolate(im, x, y, out_size)  spatial_transformer.py:200   -> output
  get_4_pts(theta, batch_size)     s_net_bundle_nobm.py:29      -> (pts1, pts2)

Tensors are torch CUDA(HIP) float32, NHWC.  No CPU fallback."""
from __future__ import annotations

import torch

from . import _lib
from ._tensor import dev_f32, empty, ptr, stream_ptr
from .config import Config, v2_93


def get_4_pts(theta: torch.Tensor, batch_size=None, cfg: Config = v2_93, with_Hs: bool = False):
    theta = dev_f32(theta, "theta")
    N = theta.shape[0]
    gh, gw = cfg.grid_h, cfg.grid_w
    assert theta.shape[1] == (gh + 1) * (gw + 1) * 2
    pts2 = empty((N, gh + 1, gw + 1, 2), theta)
    Hs = empty((N, gh, gw, 9), theta)
    pts1 = empty((N, gh, gw, 8), theta)      # per cell [xTL,xTR,xBL,xBR,yTL,yTR,yBL,yBR] (s_net_bundle_nobm.py:65-66)
    _lib.call("stabnet_get_4_pts", ptr(theta), N, gh, gw, cfg.do_crop_rate, ptr(pts1), ptr(pts2), ptr(Hs),
              stream_ptr(theta.device), device=theta.device)
    if with_Hs:
        return pts1, pts2, Hs
    return pts1, pts2


def transformer(U: torch.Tensor, theta: torch.Tensor, name="SpatialTransformer", cfg: Config = v2_93,
                return_Hs: bool = False):
    """U [N,H,W,C]; theta = pts2 [N,gh+1,gw+1,2] -> (output [N,H,W,C], black_pix [N,H,W], img [N,H,W,2])."""
    U = dev_f32(U, "U")
    pts2 = dev_f32(theta, "theta")
    N, H, W, C = U.shape
    out = empty((N, H, W, C), U)
    black = empty((N, H, W), U)
    xm = empty((N, H, W), U)
    ym = empty((N, H, W), U)
    Hs = empty((N, cfg.grid_h, cfg.grid_w, 9), U)
    _lib.call("stabnet_transformer_fwd", ptr(pts2), ptr(U), N, H, W, C, cfg.grid_h, cfg.grid_w, ptr(out), ptr(black),
              ptr(xm), ptr(ym), ptr(Hs), stream_ptr(U.device), device=U.device)
    img = empty((N, H, W, 2), U)                            # img = [x_map, y_map] (spatial_transformer3.py:295)
    _lib.call("stabnet_interleave2", ptr(xm), ptr(ym), N * H * W, ptr(img), stream_ptr(U.device), device=U.device)
    if return_Hs:
        return out, black, img, Hs
    return out, black, img


def warp_from_theta(U: torch.Tensor, theta: torch.Tensor, cfg: Config = v2_93):
    """Fused get_4_pts + transformer on the raw regressor output theta [N,50].
    -> dict(output, black_pix, x_map, y_map, Hs, pts2) with the deploy tensor shapes (deploy_bundle.py:48-56)."""
    U = dev_f32(U, "U")
    theta = dev_f32(theta, "theta")
    N, H, W, C = U.shape
    out = empty((N, H, W, C), U)
    black = empty((N, H, W), U)
    xm = empty((N, H, W, 1), U)
    ym = empty((N, H, W, 1), U)
    Hs = empty((N, cfg.grid_h, cfg.grid_w, 9), U)
    pts2 = empty((N, cfg.grid_h + 1, cfg.grid_w + 1, 2), U)
    _lib.call("stabnet_warp_fwd", ptr(theta), ptr(U), N, H, W, C, cfg.grid_h, cfg.grid_w, cfg.do_crop_rate, ptr(out),
              ptr(black), ptr(xm), ptr(ym), ptr(Hs), ptr(pts2), stream_ptr(U.device), device=U.device)
    return {"output": out, "black_pix": black, "x_map": xm, "y_map": ym, "Hs": Hs, "pts2": pts2}


def slice_channel(x: torch.Tensor, c: int) -> torch.Tensor:
    """x [..., C] -> x[..., c:c+1] as its own contiguous tensor (x_tensor[..., 12:13], s_net_bundle_nobm.py:281)."""
    x = dev_f32(x, "x")
    C = x.shape[-1]
    out = empty(tuple(x.shape[:-1]) + (1,), x)
    _lib.call("stabnet_slice_channel", ptr(x), x.numel() // C, C, int(c), ptr(out), stream_ptr(x.device), device=x.device)
    return out


def maps_from_Hs(U: torch.Tensor, Hs: torch.Tensor, cfg: Config = v2_93):
    U = dev_f32(U, "U")
    Hs = dev_f32(Hs, "Hs")
    N, H, W, C = U.shape
    out = empty((N, H, W, C), U)
    black = empty((N, H, W), U)
    xm = empty((N, H, W), U)
    ym = empty((N, H, W), U)
    _lib.call("stabnet_maps_from_hs_fwd", ptr(Hs), ptr(U), N, H, W, C, cfg.grid_h, cfg.grid_w, ptr(out), ptr(black),
              ptr(xm), ptr(ym), stream_ptr(U.device), device=U.device)
    return out, black, xm, ym


def interpolate(im: torch.Tensor, x: torch.Tensor, y: torch.Tensor, out_size=None, name="SpatialInterpolate"):
    """im [N,H,W,C]; x,y [N,H,W,1] (or [N,H,W]) normalised coords -> [N,H,W,C]."""
    im = dev_f32(im, "im")
    x = dev_f32(x, "x")
    y = dev_f32(y, "y")
    N, H, W, C = im.shape
    assert x.numel() == N * H * W and y.numel() == N * H * W
    out = empty((N, H, W, C), im)
    _lib.call("stabnet_interp_fwd", ptr(im), ptr(x), ptr(y), N, H, W, C, ptr(out), stream_ptr(im.device), device=im.device)
    return out


def warpRevBundle2(img: torch.Tensor, x_map: torch.Tensor, y_map: torch.Tensor, rate: int = 4, return_maps: bool = False):
    """deploy_bundle.py:136-146 on the device.  img uint8 [N,H,W,3] (or [H,W,3]); x_map, y_map [N,H,W(,1)] normalised."""
    squeeze = img.dim() == 3
    if squeeze:
        img = img[None]
    if not img.is_cuda or img.dtype != torch.uint8:
        raise _lib.StabnetError("warpRevBundle2: img must be a uint8 tensor on the GPU")
    img = img.contiguous()
    N, H, W, C = img.shape
    xm = dev_f32(x_map, "x_map").reshape(N, H, W)
    ym = dev_f32(y_map, "y_map").reshape(N, H, W)
    out = torch.empty_like(img)
    ws = torch.empty(2 * N * (H // rate) * (W // rate), dtype=torch.float32, device=img.device)
    px = empty((N, H, W), xm) if return_maps else None
    py = empty((N, H, W), xm) if return_maps else None
    _lib.call("stabnet_warp_rev_bundle2", ptr(img), ptr(xm), ptr(ym), N, H, W, C, rate, ptr(out), ptr(ws), ptr(px), ptr(py),
              stream_ptr(img.device), device=img.device)
    if squeeze:
        out = out[0]
    return (out, px, py) if return_maps else out


def cvt_train2img(x: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """deploy_bundle.py:75 on the device: uint8((x + 0.5) * 255), clipped; same shape as x."""
    x = dev_f32(x, "x")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    if out.dtype != torch.uint8 or not out.is_cuda or out.numel() != x.numel() or not out.is_contiguous():
        raise _lib.StabnetError("cvt_train2img: out must be a contiguous uint8 GPU tensor of x's size")
    _lib.call("stabnet_cvt_train2img", ptr(x), ptr(out), x.numel(), stream_ptr(x.device), device=x.device)
    return out


def black_accumulate(black: torch.Tensor, all_black: torch.Tensor):
    """all_black (int32, same numel) += round(black)   (deploy_bundle.py:291)."""
    black = dev_f32(black, "black")
    assert all_black.dtype == torch.int32 and all_black.is_cuda and all_black.numel() == black.numel()
    _lib.call("stabnet_black_accumulate", ptr(black), ptr(all_black), black.numel(), stream_ptr(black.device),
              device=black.device)
    return all_black


def max_inscribed_rect(all_black: torch.Tensor, step: int = 10):
    """deploy_bundle.py:344-366 on the device.  all_black int32 [H,W] -> ([i, j, hh, ww], area) as Python ints
    (([], 0) when there is no free start pixel).  Synchronises (one 20-byte read-back per video)."""
    assert all_black.dtype == torch.int32 and all_black.is_cuda and all_black.dim() == 2
    all_black = all_black.contiguous()
    H, W = all_black.shape
    nbytes = _lib.lib().stabnet_crop_search_workspace_bytes(H, W, step)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=all_black.device)
    ans = torch.empty(5, dtype=torch.int32, device=all_black.device)
    _lib.call("stabnet_crop_search", ptr(all_black), H, W, step, ptr(ans), ptr(ws), ws.numel() * 8,
              stream_ptr(all_black.device), device=all_black.device)
    a = ans.cpu().tolist()
    return (a[:4], a[4]) if a[4] > 0 else ([], 0)
