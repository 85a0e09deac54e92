"""Thin host wrappers of the regressor building blocks of the C ABI (used by tests and by regressor.py)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._tensor import dev_f32, empty, ptr, stream_ptr


def pack_conv_weight(w_hwio, cin_pad: int = 16) -> np.ndarray:
    """TF HWIO [kh,kw,Cin,Cout] -> the library's OHWI [Cout,kh,kw,Cin'] with Cin' = Cin rounded up to `cin_pad`."""
    w = np.asarray(w_hwio, np.float32)
    kh, kw, ci, co = w.shape
    cp = -(-ci // cin_pad) * cin_pad
    out = np.zeros((co, kh, kw, cp), np.float32)
    out[..., :ci] = np.transpose(w, (3, 0, 1, 2))
    return out


def unpack_conv_weight(w_ohwi, cin: int) -> np.ndarray:
    return np.ascontiguousarray(np.transpose(np.asarray(w_ohwi, np.float32)[..., :cin], (1, 2, 3, 0)))


def conv2d(x, w_ohwi, bias=None, in_scale=None, in_shift=None, residual=None, res_stride=1, stride=1, pad=0,
           relu_out=False, out_scale=None, out_shift=None):
    x = dev_f32(x, "x")
    w = dev_f32(w_ohwi, "w")
    N, H, W, Cin = x.shape
    Cout, KH, KW, Cw = w.shape
    assert Cw == Cin, "weight Cin %d != input Cin %d" % (Cw, Cin)
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    y = empty((N, Ho, Wo, Cout), x)
    L = _lib.lib()
    ws_bytes = L.stabnet_conv2d_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad)
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=x.device)
    rH, rW = (residual.shape[1], residual.shape[2]) if residual is not None else (0, 0)
    _lib.call("stabnet_conv2d_fwd_ex", ptr(x), ptr(w), ptr(bias), ptr(in_scale), ptr(in_shift), ptr(residual), rH, rW,
              res_stride, ptr(out_scale), ptr(out_shift), ptr(y), N, H, W, Cin, Cout, KH, KW, stride, pad,
              int(relu_out), ptr(ws), ws_bytes, stream_ptr(x.device), device=x.device)
    return y


def conv2d_packed(x, w_ohwi, bias=None, in_scale=None, in_shift=None, residual=None, res_stride=1, stride=1, pad=0,
                  relu_out=False, out_scale=None, out_shift=None, splitk=0):
    """conv2d on the packed split kernels (stabnet_conv2d_fwd_packed): float32-level results on the bf16 matrix pipe."""
    x = dev_f32(x, "x")
    w = dev_f32(w_ohwi, "w")
    N, H, W, Cin = x.shape
    Cout, KH, KW, Cw = w.shape
    assert Cw == Cin, "weight Cin %d != input Cin %d" % (Cw, Cin)
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    y = empty((N, Ho, Wo, Cout), x)
    L = _lib.lib()
    n_img = int(L.stabnet_conv_weight_image_floats(Cout, KH, KW, Cin))
    img = torch.empty(max(n_img, 1), dtype=torch.float32, device=x.device)
    if n_img > 0:                      # (Cin not a multiple of 32: no image exists and the call runs the exact-f32 kernels on w_ohwi)
        _lib.call("stabnet_conv_weight_split_image", ptr(w), Cout, KH, KW, Cin, ptr(img), stream_ptr(x.device), device=x.device)
    ws_bytes = max(int(L.stabnet_conv2d_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad)), max(splitk, 0) * N * Ho * Wo * Cout * 4)
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=x.device)
    rH, rW = (residual.shape[1], residual.shape[2]) if residual is not None else (0, 0)
    _lib.call("stabnet_conv2d_fwd_packed", ptr(x), ptr(w), ptr(img), ptr(bias), ptr(in_scale), ptr(in_shift), ptr(residual), rH, rW,
              res_stride, ptr(out_scale), ptr(out_shift), ptr(y), N, H, W, Cin, Cout, KH, KW, stride, pad, int(relu_out), int(splitk),
              ptr(ws), ws_bytes, stream_ptr(x.device), device=x.device)
    return y


def conv3x3_conv1x1(x, w2_ohwi, mid_scale, mid_shift, w3_ohwi, bias3=None, residual=None, res_stride=1, stride=1, relu_out=False,
                    out_scale=None, out_shift=None, x_ch0=0, res_ch0=0):
    """The tail of a bottleneck unit as one launch (stabnet_conv3x3_conv1x1_fwd): conv2 3x3 (pad 1, `stride`) -> folded BN + ReLU
    -> conv3 1x1 with conv2d's epilogue.  x [N,H,W,Cx]: the conv reads channels x_ch0 .. x_ch0 + C of every pixel (Cx > C: the
    inference plan's merged shortcut|conv1 buffer); residual [N,rH,rW,Cr]: channels res_ch0 .. res_ch0 + Cout likewise."""
    x = dev_f32(x, "x")
    w2 = dev_f32(w2_ohwi, "w2")
    w3 = dev_f32(w3_ohwi, "w3")
    N, H, W, Cx = x.shape
    C = w2.shape[0]
    Cout = w3.shape[0]
    assert tuple(w2.shape) == (C, 3, 3, C) and tuple(w3.shape[1:]) == (1, 1, C) and x_ch0 + C <= Cx
    Ho = (H + 2 - 3) // stride + 1
    Wo = (W + 2 - 3) // stride + 1
    y = empty((N, Ho, Wo, Cout), x)
    rH = rW = res_ld = 0
    rp = 0
    if residual is not None:
        residual = dev_f32(residual, "residual")
        rH, rW, res_ld = residual.shape[1], residual.shape[2], residual.shape[3]
        assert res_ch0 + Cout <= res_ld
        rp = residual.data_ptr() + 4 * res_ch0
    _lib.call("stabnet_conv3x3_conv1x1_fwd", x.data_ptr() + 4 * x_ch0, Cx, ptr(w2), ptr(mid_scale), ptr(mid_shift), ptr(w3),
              ptr(bias3), rp, rH, rW, res_stride, res_ld, ptr(out_scale), ptr(out_shift), ptr(y), N, H, W, C, Cout, stride,
              int(relu_out), stream_ptr(x.device), device=x.device)
    return y


def conv2d_wgrad(x, dy, w_shape, in_scale=None, in_shift=None, stride=1, pad=0, dw=None):
    """dW (OHWI) += d conv / d W; returns dw (zero-initialised when not given)."""
    x = dev_f32(x, "x")
    dy = dev_f32(dy, "dy")
    N, H, W, Cin = x.shape
    Cout, KH, KW, Cw = w_shape
    assert Cw == Cin
    if dw is None:
        dw = torch.zeros(w_shape, dtype=torch.float32, device=x.device)
    nbytes = _lib.lib().stabnet_conv2d_wgrad_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    _lib.call("stabnet_conv2d_wgrad", ptr(x), ptr(dy), ptr(dw), ptr(in_scale), ptr(in_shift), N, H, W, Cin, Cout, KH, KW,
              stride, pad, ptr(ws), nbytes, stream_ptr(x.device), device=x.device)
    return dw


def conv2d_wgrad_bias(x, dy, w_shape, in_scale=None, in_shift=None, dw=None, db=None):
    """1x1 layers: (dW, d_bias) += gradients, the bias sums taken from the wgrad kernel's own pass over dy."""
    x = dev_f32(x, "x")
    dy = dev_f32(dy, "dy")
    N, H, W, Cin = x.shape
    Cout = w_shape[0]
    assert tuple(w_shape[1:]) == (1, 1, Cin)
    buf = torch.zeros(Cout * Cin + Cout, dtype=torch.float32, device=x.device)      # one allocation: d_bias sits behind dW
    if dw is not None:
        buf[:Cout * Cin] = dw.reshape(-1)
    if db is not None:
        buf[Cout * Cin:] = db
    nbytes = _lib.lib().stabnet_conv2d_wgrad_bias_workspace_bytes(N, H, W, Cin, Cout)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    _lib.call("stabnet_conv2d_wgrad_bias", ptr(x), ptr(dy), buf.data_ptr(), buf.data_ptr() + 4 * Cout * Cin, ptr(in_scale), ptr(in_shift),
              N, H, W, Cin, Cout, ptr(ws), nbytes, stream_ptr(x.device), device=x.device)
    return buf[:Cout * Cin].reshape(w_shape), buf[Cout * Cin:]


def conv2d_wgrad_rowrun(x, dy, w_shape, stride=1, pad=0, dw=None):
    """dW (OHWI [Cout,KH,KW,CinPad], CinPad >= Cin) += d conv / d W for an input whose channel count is not a multiple of 4
    (the 13-channel stem): the filter-row-run form the training step uses.  Pad channels of dw are not written."""
    x = dev_f32(x, "x")
    dy = dev_f32(dy, "dy")
    N, H, W, Cin = x.shape
    Cout, KH, KW, CinPad = w_shape
    if dw is None:
        dw = torch.zeros(w_shape, dtype=torch.float32, device=x.device)
    nbytes = _lib.lib().stabnet_conv2d_wgrad_rowrun_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    _lib.call("stabnet_conv2d_wgrad_rowrun", ptr(x), ptr(dy), ptr(dw), N, H, W, Cin, CinPad, Cout, KH, KW, stride, pad,
              ptr(ws), nbytes, stream_ptr(x.device), device=x.device)
    return dw


def conv2d_dgrad(dy, w_ohwi, x_shape, stride=1, pad=0, residual=None):
    dy = dev_f32(dy, "dy")
    w = dev_f32(w_ohwi, "w")
    N, H, W, Cin = x_shape
    Cout, KH, KW, _ = w.shape
    dx = torch.empty(x_shape, dtype=torch.float32, device=dy.device)
    L = _lib.lib()
    nbytes = L.stabnet_conv2d_dgrad_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    _lib.call("stabnet_conv2d_dgrad", ptr(dy), ptr(w), ptr(dx), ptr(residual), N, H, W, Cin, Cout, KH, KW, stride, pad,
              ptr(ws), nbytes, stream_ptr(dy.device), device=dy.device)
    return dx
