"""CPU oracle for the StabNet hot path (NumPy, float32 arithmetic op-for-op).

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file; only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and
only as the checker.

PARITY UNPINNED: the reference (TensorFlow 1.3 + tf.contrib.slim + cv2) cannot be
imported in the build container (modules absent, no network) and ships no tests,
golden vectors or fixtures for this path.  This file is a
restatement made by reading the reference sources as text; every function cites the
reference file:line it follows.  Third-party arithmetic that is not under
/root/reference (pinned only by README.md:8 `tensorflow-gpu==1.3.0`) is restated from
its published algorithm and flagged `[external]`:
  * tf.linspace (LinSpaceOp CPU kernel): `start + step * i`, step = (stop-start)/(n-1)
  * tf.matrix_inverse -> Eigen 3.3 PartialPivLU::inverse (unblocked LU for n <= 16,
    column-oriented triangular solves multiplying by the reciprocal diagonal)
  * tf.contrib.slim.nets.resnet_v2.resnet_v2_50 / batch_norm / max_pool2d / fully_connected
  * tf.round = round-half-to-even; float->int32 cast with x86 cvttss2si semantics

All arithmetic is float32 with one rounding per TF op (TF 1.3 CPU wheels are built
without FMA, and every elementwise TF op is its own kernel), so the HIP warp path can
be compared bit-for-bit.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

F = np.float32
INT_MIN = np.int32(-2147483648)


# --------------------------------------------------------------------------- config
@dataclass
class Config:
    """Constants of configs/v2_93.py:3-49 (only those the hot path reads)."""
    height: int = 288
    width: int = 512
    batch_size: int = 10
    grid_h: int = 4
    grid_w: int = 4
    before_ch: int = 6
    tot_ch: int = 7
    input_mask: bool = True
    do_crop_rate: float = 0.8
    max_matches: int = 3000
    feature_mul: float = 1
    theta_mul: float = 400 / 2500
    regu_mul: float = 30 / 2500
    img_mul: float = 50
    temp_mul: float = 500
    black_mul: float = 300000 / 2500
    id_mul: float = 10 / 2500
    distortion_mul: float = 1
    consistency_mul: float = 20
    grid_theta_mul: float = 0
    weight_decay_fc: float = 0.0002      # hyper_parameters.py:38 via resnet.py:35-37
    weight_decay_conv: float = 0.0001    # [external] slim resnet_arg_scope default
    bn_eps: float = 1e-5                 # [external] slim resnet_arg_scope default
    bn_decay: float = 0.997              # [external]
    indices: tuple = (0, 1, 2, 4, 8, 16, 32)

    @property
    def in_ch(self):
        return self.tot_ch + self.before_ch if self.input_mask else self.tot_ch


def _f(x):
    return np.asarray(x, dtype=F)


# --------------------------------------------------------------------------- mesh
def get_4_pts(theta, cfg: Config):
    """s_net_bundle_nobm.py:29-71.  theta [N,(gh+1)(gw+1)2] -> pts1 [N,gh,gw,8], pts2 [N,gh+1,gw+1,2].

    vertex(i,j) = (j*w-1, i*h-1) + theta[2*(i*(gw+1)+j) : +2]  (x first, :46-47,:55),
    clipped to +-1/do_crop_rate (:37,:58).  pts1 per cell = [xTL,xTR,xBL,xBR,yTL,yTR,yBL,yBR]
    (:65-66: concat of [N,2,1] columns along axis 2, then flatten).
    """
    theta = _f(theta)
    N = theta.shape[0]
    gh, gw = cfg.grid_h, cfg.grid_w
    h = 2.0 / gh
    w = 2.0 / gw
    lim = F(1.0) / F(cfg.do_crop_rate)
    pts2 = np.empty((N, gh + 1, gw + 1, 2), F)
    tot = 0
    for i in range(gh + 1):
        for j in range(gw + 1):
            base = np.array([j * w - 1, i * h - 1], dtype=F)
            p = base[None, :] + theta[:, 2 * tot:2 * tot + 2]
            p = np.minimum(np.maximum(p, -lim), lim)
            pts2[:, i, j, :] = p
            tot += 1
    pts1 = np.empty((N, gh, gw, 8), F)
    for i in range(gh):
        for j in range(gw):
            g = np.stack([pts2[:, i, j], pts2[:, i, j + 1], pts2[:, i + 1, j], pts2[:, i + 1, j + 1]], axis=2)
            pts1[:, i, j, :] = g.reshape(N, 8)          # [N,2,4] -> x*4, y*4
    return pts1, pts2


# --------------------------------------------------------------------------- 8x8 inverse
def inv8_partial_piv_lu(A):
    """[external] tf.matrix_inverse == Eigen::PartialPivLU<Matrix>(A).inverse(), float32.

    Restated for n = 8 (< Eigen's UnBlockedBound 16, so the unblocked right-looking LU runs):
      for k: pivot = first argmax |a[k:,k]|; swap rows; a[k+1:,k] /= a[k,k];
             a[k+1:,k+1:] -= a[k+1:,k] (x) a[k,k+1:]                       (mul then sub)
      X = P*I; unit-lower solve then upper solve, both column oriented:
             b = x[i,:] *= 1/u[i,i] (upper only);  x[rest,:] -= b * tri[rest,i]
    A: [B,8,8] float32 -> inverse [B,8,8] float32.  Every op is a separate float32 rounding.
    """
    A = np.array(A, dtype=F, copy=True)
    B, n, _ = A.shape
    bi = np.arange(B)
    perm = np.tile(np.arange(n), (B, 1))
    for k in range(n):
        piv = np.argmax(np.abs(A[:, k:, k]), axis=1) + k
        rk = A[bi, k, :].copy()
        rp = A[bi, piv, :].copy()
        A[bi, k, :] = rp
        A[bi, piv, :] = rk
        pk = perm[bi, k].copy()
        perm[bi, k] = perm[bi, piv]
        perm[bi, piv] = pk
        if k + 1 < n:
            A[:, k + 1:, k] = A[:, k + 1:, k] / A[:, k, k][:, None]
            prod = A[:, k + 1:, k][:, :, None] * A[:, k, k + 1:][:, None, :]
            A[:, k + 1:, k + 1:] = A[:, k + 1:, k + 1:] - prod
    X = np.zeros((B, n, n), F)
    X[bi[:, None], np.arange(n)[None, :], perm] = F(1.0)      # X = P * I : row i has its 1 in column perm[i]
    for i in range(n):                                         # unit-lower, ascending
        b = X[:, i, :]
        if i + 1 < n:
            X[:, i + 1:, :] = X[:, i + 1:, :] - b[:, None, :] * A[:, i + 1:, i][:, :, None]
    for i in range(n - 1, -1, -1):                             # upper, descending
        a = F(1.0) / A[:, i, i]
        X[:, i, :] = X[:, i, :] * a[:, None]
        b = X[:, i, :]
        if i > 0:
            X[:, :i, :] = X[:, :i, :] - b[:, None, :] * A[:, :i, i][:, :, None]
    return X


def get_H(ori, tar):
    """spatial_transformer3.py:144-175.  ori,tar [B,8] interleaved (x0,y0,...,x3,y3) -> [B,9].

    A rows: 4 u-rows [x,y,1,0,0,0,-x*u,-y*u] then 4 v-rows [0,0,0,x,y,1,-x*v,-y*v] (:160-167);
    b = [u0..u3,v0..v3] (:169); h = inv(A + eye(8)*1e-4) @ b (:145,:173); append 1.
    """
    ori = _f(ori)
    tar = _f(tar)
    B = ori.shape[0]
    x, y = ori[:, 0::2], ori[:, 1::2]
    u, v = tar[:, 0::2], tar[:, 1::2]
    A = np.zeros((B, 8, 8), F)
    for r in range(4):
        A[:, r, 0] = x[:, r]
        A[:, r, 1] = y[:, r]
        A[:, r, 2] = 1
        A[:, r, 6] = (-x[:, r]) * u[:, r]
        A[:, r, 7] = (-y[:, r]) * u[:, r]
        A[:, 4 + r, 3] = x[:, r]
        A[:, 4 + r, 4] = y[:, r]
        A[:, 4 + r, 5] = 1
        A[:, 4 + r, 6] = (-x[:, r]) * v[:, r]
        A[:, 4 + r, 7] = (-y[:, r]) * v[:, r]
    b = np.concatenate([u, v], axis=1)
    ridge = np.eye(8, dtype=F) * F(1e-4)
    Ainv = inv8_partial_piv_lu(A + ridge[None])
    h = np.zeros((B, 8), F)
    for k in range(8):                                    # sequential k-sum of the batched matmul
        h = h + Ainv[:, :, k] * b[:, k][:, None]
    return np.concatenate([h, np.ones((B, 1), F)], axis=1)


def get_Hs(pts2, cfg: Config):
    """spatial_transformer3.py:179-198.  pts2 [N,gh+1,gw+1,2] -> Hs [N,gh,gw,9]."""
    pts2 = _f(pts2)
    N = pts2.shape[0]
    gh, gw = cfg.grid_h, cfg.grid_w
    h = 2.0 / gh
    w = 2.0 / gw
    Hs = np.empty((N, gh, gw, 9), F)
    for i in range(gh):
        for j in range(gw):
            hh = i * h - 1
            ww = j * w - 1
            ori = np.tile(np.array([ww, hh, ww + w, hh, ww, hh + h, ww + w, hh + h], dtype=F)[None], (N, 1))
            tar = np.concatenate([pts2[:, i, j], pts2[:, i, j + 1], pts2[:, i + 1, j], pts2[:, i + 1, j + 1]], axis=1)
            Hs[:, i, j] = get_H(ori, tar)
    return Hs


def linspace_tf(start, stop, num):
    """[external] TF 1.x LinSpaceOp: flat(i) = start + step * i, step = (stop - start)/(num - 1), in float32."""
    start, stop = F(start), F(stop)
    if num == 1:
        return np.array([start], F)
    step = (stop - start) / F(num - 1)
    return start + step * np.arange(num, dtype=F)


def cast_i32_x86(x):
    """[external] tf.cast(float32 -> int32) on x86 (cvttss2si): out-of-range / NaN -> INT_MIN."""
    x = _f(x)
    ok = (x >= F(-2147483648.0)) & (x < F(2147483648.0))
    return np.where(ok, np.where(ok, x, F(0)).astype(np.int32), INT_MIN).astype(np.int32)


# --------------------------------------------------------------------------- sampler
def _interpolate(im, x, y):
    """spatial_transformer3.py:62-123 (identical copy spatial_transformer.py:209-270).

    im [N,H,W,C]; x,y flat [N*H*W] normalised coords -> [N*H*W, C].
    Corners are clipped BEFORE the weights are formed (:90-93,:114-121).
    """
    im = _f(im)
    N, H, W, C = im.shape
    x = _f(x)
    y = _f(y)
    xp = (x + F(1.0)) * F(W) / F(2.0)
    yp = (y + F(1.0)) * F(H) / F(2.0)
    x0 = cast_i32_x86(np.floor(xp))
    x1 = x0 + np.int32(1)
    y0 = cast_i32_x86(np.floor(yp))
    y1 = y0 + np.int32(1)
    x0 = np.clip(x0, 0, W - 1)
    x1 = np.clip(x1, 0, W - 1)
    y0 = np.clip(y0, 0, H - 1)
    y1 = np.clip(y1, 0, H - 1)
    base = np.repeat(np.arange(N, dtype=np.int64) * (H * W), x.size // N)
    idx_a = base + y0.astype(np.int64) * W + x0
    idx_b = base + y1.astype(np.int64) * W + x0
    idx_c = base + y0.astype(np.int64) * W + x1
    idx_d = base + y1.astype(np.int64) * W + x1
    flat = im.reshape(-1, C)
    Ia, Ib, Ic, Id = flat[idx_a], flat[idx_b], flat[idx_c], flat[idx_d]
    x0f, x1f, y0f, y1f = x0.astype(F), x1.astype(F), y0.astype(F), y1.astype(F)
    wa = ((x1f - xp) * (y1f - yp))[:, None]
    wb = ((x1f - xp) * (yp - y0f))[:, None]
    wc = ((xp - x0f) * (y1f - yp))[:, None]
    wd = ((xp - x0f) * (yp - y0f))[:, None]
    out = ((wa * Ia + wb * Ib) + wc * Ic) + wd * Id           # tf.add_n, left to right
    return out, (x0, y0, x1, y1)


def interpolate(im, x, y, out_size=None):
    """spatial_transformer.py:200-281.  im [N,H,W,C]; x,y [N,H,W,1] -> [N,H,W,C]."""
    im = _f(im)
    N, H, W, C = im.shape
    out, _ = _interpolate(im, _f(x).reshape(-1), _f(y).reshape(-1))
    return out.reshape(N, H, W, C)


# --------------------------------------------------------------------------- warp
def cell_bounds(H, W, gh, gw):
    """Pixel ownership of the cells, spatial_transformer3.py:227-243."""
    ch = int(math.floor(H / gh))
    cw = int(math.floor(W / gw))
    rows = [(i * ch, (i + 1) * ch - 1 if i < gh - 1 else H - 1) for i in range(gh)]
    cols = [(j * cw, (j + 1) * cw - 1 if j < gw - 1 else W - 1) for j in range(gw)]
    return rows, cols


def maps_from_Hs(Hs, H, W, cfg: Config):
    """spatial_transformer3.py:200-214,227-286: per-cell H . (x,y,1), z += sign*1e-8, divide; black test."""
    Hs = _f(Hs)
    N = Hs.shape[0]
    gh, gw = cfg.grid_h, cfg.grid_w
    xs = linspace_tf(-1.0, 1.0, W)
    ys = linspace_tf(-1.0, 1.0, H)
    rows, cols = cell_bounds(H, W, gh, gw)
    x_map = np.empty((N, H, W), F)
    y_map = np.empty((N, H, W), F)
    one = F(1.0)
    for i, (sh, eh) in enumerate(rows):
        for j, (sw, ew) in enumerate(cols):
            h = Hs[:, i, j, :][:, :, None, None]                       # [N,9,1,1]
            gx = xs[sw:ew + 1][None, None, :]
            gy = ys[sh:eh + 1][None, :, None]
            tx = (h[:, 0] * gx + h[:, 1] * gy) + h[:, 2] * one        # k-sequential matmul, :248
            ty = (h[:, 3] * gx + h[:, 4] * gy) + h[:, 5] * one
            tz = (h[:, 6] * gx + h[:, 7] * gy) + h[:, 8] * one
            sign = np.where(tz >= 0, one, F(0.0)) * F(2.0) - one       # :257
            tz = tz + sign * F(1e-8)                                   # :258
            x_map[:, sh:eh + 1, sw:ew + 1] = tx / tz
            y_map[:, sh:eh + 1, sw:ew + 1] = ty / tz
    black = ((-one > x_map) | (x_map > one) | (-one > y_map) | (y_map > one)).astype(F)   # :284-286
    return x_map, y_map, black


def transformer(U, pts2, cfg: Config, return_all=False):
    """spatial_transformer3.py:19,218-301,362-365.

    U [N,H,W,C], pts2 [N,gh+1,gw+1,2] -> (output [N,H,W,C], black_pix [N,H,W], img [N,H,W,2]).
    """
    U = _f(U)
    N, H, W, C = U.shape
    Hs = get_Hs(pts2, cfg)
    x_map, y_map, black = maps_from_Hs(Hs, H, W, cfg)
    out, corners = _interpolate(U, x_map.reshape(-1), y_map.reshape(-1))
    out = out.reshape(N, H, W, C)
    img = np.stack([x_map, y_map], axis=3)
    if return_all:
        return out, black, img, Hs, corners
    return out, black, img


# --------------------------------------------------------------------------- losses
def get_black_pos(pts1, cfg: Config):
    """s_net_bundle_nobm.py:139-146: hinge of pts1 beyond +-1/do_crop_rate."""
    pts1 = _f(pts1)
    lim = F(1.0) / F(cfg.do_crop_rate)
    z = F(0.0)
    err = np.where(pts1 > lim, pts1 - lim, z) + np.where(-lim > pts1, -lim - pts1, z)
    return err.reshape(pts1.shape[0], -1)


def _calc_distortion(p0, p1, p2, clock, hw, cfg):
    """s_net_bundle_nobm.py:148-164."""
    h = 2.0 / cfg.grid_h
    w = 2.0 / cfg.grid_w
    k = h / w if hw == 0 else w / h
    R = np.array([0, -k, k, 0] if not clock else [0, k, -k, 0], dtype=F).reshape(2, 2)
    d = p1 - p0
    rd = np.stack([R[0, 0] * d[:, 0] + R[0, 1] * d[:, 1], R[1, 0] * d[:, 0] + R[1, 1] * d[:, 1]], axis=1)
    loss = np.abs(rd - (p2 - p1))
    return loss * loss


def get_distortion_loss(pts1, cfg: Config):
    """s_net_bundle_nobm.py:166-181."""
    pts = _f(pts1).reshape(-1, 2, 4)
    p0, p1, p2, p3 = pts[:, :, 0], pts[:, :, 1], pts[:, :, 2], pts[:, :, 3]
    loss = _calc_distortion(p0, p1, p3, 0, 0, cfg)
    loss = loss + _calc_distortion(p1, p3, p2, 0, 1, cfg)
    loss = loss + _calc_distortion(p3, p2, p0, 0, 0, cfg)
    loss = loss + _calc_distortion(p2, p0, p1, 0, 1, cfg)
    loss = loss + _calc_distortion(p1, p0, p2, 1, 0, cfg)
    loss = loss + _calc_distortion(p0, p2, p3, 1, 1, cfg)
    loss = loss + _calc_distortion(p2, p3, p1, 1, 0, cfg)
    loss = loss + _calc_distortion(p3, p1, p0, 1, 1, cfg)
    return F(np.mean(loss, dtype=np.float64)) / F(8)


def get_consistency_loss(pts2, cfg: Config):
    """s_net_bundle_nobm.py:183-210."""
    p = _f(pts2)
    gh, gw = cfg.grid_h, cfg.grid_w
    errs = []
    two = F(2.0)
    for i in range(gh + 1):
        for j in range(gw + 1):
            if i > 1:
                errs.append(np.abs(two * p[:, i - 1, j] - p[:, i, j] - p[:, i - 2, j]))
            if j > 1:
                errs.append(np.abs(two * p[:, i, j - 1] - p[:, i, j] - p[:, i, j - 2]))
            if i < gh - 1:
                errs.append(np.abs(two * p[:, i + 1, j] - p[:, i, j] - p[:, i + 2, j]))
            if j < gw - 1:
                errs.append(np.abs(two * p[:, i, j + 1] - p[:, i, j] - p[:, i, j + 2]))
    if not errs:
        return F(0.0)
    e = np.stack(errs, axis=2)
    return F(np.mean(e * e, dtype=np.float64))


def warp_pts(pts, flow, cfg: Config):
    """s_net_bundle_nobm.py:215-230.  pts [N,M,2] normalised, flow(maps) [N,H,W,2] -> [N,M,2]."""
    pts = _f(pts)
    flow = _f(flow)
    N, H, W, _ = flow.shape
    x = np.clip((pts[:, :, 0] + F(1)) / F(2) * F(W), F(0), F(W - 1))
    y = np.clip((pts[:, :, 1] + F(1)) / F(2) * F(H), F(0), F(H - 1))
    xi = np.rint(x).astype(np.int32)            # tf.round: half to even
    yi = np.rint(y).astype(np.int32)
    out = np.empty((N, pts.shape[1], 2), F)
    for n in range(N):
        out[n] = flow[n].reshape(-1, 2)[xi[n] + yi[n] * W]
    return out, (xi, yi)


def feature_loss(matches, mask, flow, cfg: Config):
    """s_net_bundle_nobm.py:335-343."""
    matches = _f(matches)
    mask = _f(mask)
    stable, unstable = matches[:, :, :2], matches[:, :, 2:]
    warped, _ = warp_pts(stable, flow, cfg)
    before = np.sum(np.abs(warped - unstable), axis=2, dtype=F)
    after = np.sum(before * mask, axis=1, dtype=np.float64) / np.maximum(np.sum(mask, axis=1, dtype=np.float64), 1.0)
    return F(np.mean(after)), warped


def img_loss(h_trans, y, black_pix, cfg: Config):
    """s_net_bundle_nobm.py:347-352 (divides by the STATIC batch_size)."""
    h_trans = _f(h_trans)
    N = h_trans.shape[0]
    keep = F(1) - _f(black_pix).reshape(N, h_trans.shape[1], h_trans.shape[2], 1)
    err = (h_trans - _f(y)) * keep
    num = np.sum(err * err, axis=(1, 2, 3), dtype=np.float64)
    den = np.sum(keep, axis=(1, 2, 3), dtype=np.float64) + 1e-8
    return F(np.sum(num / den) / cfg.batch_size)


def temporal_loss(out1, black1, out2, black2, flow, cfg: Config, use_temp_loss=1.0):
    """train_bundle_nobm.py:110-125."""
    out1, out2 = _f(out1), _f(out2)
    N, H, W, _ = out1.shape
    fx, fy = _f(flow)[..., 0:1], _f(flow)[..., 1:2]
    o2 = interpolate(out2, fx, fy)
    nb2 = interpolate((F(1) - _f(black2)).reshape(N, H, W, 1), fx, fy)
    err = out1 - o2
    noblack = (F(1) - _f(black1)).reshape(N, H, W, 1) * nb2
    err = err * noblack
    num = np.sum(err * err, axis=(1, 2, 3), dtype=np.float64)
    den = np.sum(noblack, axis=(1, 2, 3), dtype=np.float64) + 1e-8
    return F(np.sum(num / den) / cfg.batch_size * use_temp_loss)


# --------------------------------------------------------------------------- backbone  [external] slim
def _same_pads(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return tot // 2, tot - tot // 2


def conv2d(x, w, stride=1, pads=((0, 0), (0, 0)), bias=None):
    """NHWC x, HWIO w, explicit zero pads, VALID after padding; im2col + one sgemm."""
    x = _f(x)
    w = _f(w)
    kh, kw, ci, co = w.shape
    if pads != ((0, 0), (0, 0)):
        x = np.pad(x, ((0, 0), pads[0], pads[1], (0, 0)))
    N, H, W, _ = x.shape
    Ho = (H - kh) // stride + 1
    Wo = (W - kw) // stride + 1
    if kh == 1 and kw == 1:
        cols = x[:, ::stride, ::stride, :][:, :Ho, :Wo, :].reshape(-1, ci)
    else:
        s0, s1, s2, s3 = x.strides
        v = np.lib.stride_tricks.as_strided(
            x, (N, Ho, Wo, kh, kw, ci), (s0, s1 * stride, s2 * stride, s1, s2, s3), writeable=False)
        cols = np.ascontiguousarray(v).reshape(N * Ho * Wo, kh * kw * ci)
    y = cols @ w.reshape(kh * kw * ci, co)
    if bias is not None:
        y = y + _f(bias)[None, :]
    return y.reshape(N, Ho, Wo, co)


def conv2d_same(x, w, stride, bias=None):
    """[external] slim resnet_utils.conv2d_same: stride 1 -> SAME; else explicit symmetric-ish pad + VALID."""
    k = w.shape[0]
    if stride == 1:
        ph = _same_pads(x.shape[1], k, 1)
        pw = _same_pads(x.shape[2], k, 1)
        return conv2d(x, w, 1, (ph, pw), bias)
    tot = k - 1
    beg = tot // 2
    end = tot - beg
    return conv2d(x, w, stride, ((beg, end), (beg, end)), bias)


def max_pool_3x3_s2_same(x):
    """[external] slim max_pool2d(3, stride 2, padding='SAME'): TF-SAME pads (before=tot//2, after=rest) with -inf."""
    x = _f(x)
    ph = _same_pads(x.shape[1], 3, 2)
    pw = _same_pads(x.shape[2], 3, 2)
    xp = np.pad(x, ((0, 0), ph, pw, (0, 0)), constant_values=-np.inf)
    N, H, W, C = xp.shape
    Ho = (H - 3) // 2 + 1
    Wo = (W - 3) // 2 + 1
    out = np.full((N, Ho, Wo, C), -np.inf, F)
    for dy in range(3):
        for dx in range(3):
            out = np.maximum(out, xp[:, dy:dy + 2 * Ho:2, dx:dx + 2 * Wo:2, :][:, :Ho, :Wo, :])
    return out


def batch_norm(x, p, prefix, cfg: Config, training=False, stats_out=None):
    """[external] slim batch_norm (non-fused, TF 1.3): tf.nn.batch_normalization:
    inv = rsqrt(var+eps)*gamma;  y = x*inv + (beta - mean*inv).
    training=True uses tf.nn.moments batch statistics (biased variance)."""
    x = _f(x)
    gamma, beta = _f(p[prefix + '/gamma']), _f(p[prefix + '/beta'])
    if training:
        x64 = x.astype(np.float64)
        mean = x64.mean(axis=(0, 1, 2))
        var = ((x64 - mean) ** 2).mean(axis=(0, 1, 2))
        mean, var = mean.astype(F), var.astype(F)
        if stats_out is not None:
            stats_out[prefix] = (mean, var)
    else:
        mean, var = _f(p[prefix + '/moving_mean']), _f(p[prefix + '/moving_variance'])
    inv = (F(1.0) / np.sqrt(var + F(cfg.bn_eps))) * gamma
    return x * inv + (beta - mean * inv)


def relu(x):
    return np.maximum(x, F(0))


RESNET_V2_50_BLOCKS = (      # (depth, bottleneck depth, units, stride on the LAST unit)  [external] slim resnet_v2_50
    ('block1', 256, 64, 3, 2),
    ('block2', 512, 128, 4, 2),
    ('block3', 1024, 256, 6, 2),
    ('block4', 2048, 512, 3, 1),
)


def resnet_v2_50(x, p, cfg: Config, training=False, stats_out=None, taps=None):
    """[external] slim resnet_v2_50(x, global_pool=False, output_stride=32) under resnet_arg_scope(),
    as called at s_net_bundle_nobm.py:252-253.  p: name -> array, TF layouts."""
    R = 'resnet_v2_50/'
    net = conv2d_same(x, p[R + 'conv1/weights'], 2, p[R + 'conv1/biases'])          # bias, no BN/ReLU
    if taps is not None:
        taps['conv1'] = net
    net = max_pool_3x3_s2_same(net)
    if taps is not None:
        taps['pool1'] = net
    for (bname, depth, dbn, units, bstride) in RESNET_V2_50_BLOCKS:
        for u in range(1, units + 1):
            stride = bstride if u == units else 1
            S = R + '%s/unit_%d/bottleneck_v2/' % (bname, u)
            depth_in = net.shape[3]
            preact = relu(batch_norm(net, p, S + 'preact', cfg, training, stats_out))
            if depth == depth_in:
                shortcut = net if stride == 1 else net[:, ::stride, ::stride, :]      # max_pool 1x1 stride s
            else:
                shortcut = conv2d(preact, p[S + 'shortcut/weights'], stride, bias=p[S + 'shortcut/biases'])
            r = conv2d(preact, p[S + 'conv1/weights'], 1)
            r = relu(batch_norm(r, p, S + 'conv1/BatchNorm', cfg, training, stats_out))
            r = conv2d_same(r, p[S + 'conv2/weights'], stride)
            r = relu(batch_norm(r, p, S + 'conv2/BatchNorm', cfg, training, stats_out))
            r = conv2d(r, p[S + 'conv3/weights'], 1, bias=p[S + 'conv3/biases'])
            net = shortcut + r
            if taps is not None:
                taps['%s/unit_%d' % (bname, u)] = net
    net = relu(batch_norm(net, p, R + 'postnorm', cfg, training, stats_out))
    return net


def get_resnet(x_tensor, p, cfg: Config, training=False, stats_out=None, taps=None):
    """s_net_bundle_nobm.py:250-264: backbone -> reduce_mean(1,2) -> FC 2048/1024/512 (ReLU) -> output_layer.
    Returns (theta [N,(gh+1)(gw+1)2], id_loss, id2_loss)."""
    feat = resnet_v2_50(x_tensor, p, cfg, training, stats_out, taps)
    g = feat.mean(axis=(1, 2), dtype=np.float64).astype(F)
    if taps is not None:
        taps['global_pool'] = g
    for k in (1, 2, 3):
        g = relu(g @ _f(p['fc/fc/fc_%d/weights' % k]) + _f(p['fc/fc/fc_%d/biases' % k]))
    theta = g @ _f(p['fc/fc_weights']) + _f(p['fc/fc_bias'])                       # resnet.py:44-56
    id2 = F(np.mean(np.abs(theta), dtype=np.float64)) * F(cfg.id_mul)              # :263
    return theta, id2, id2


def regu_loss(p, cfg: Config):
    """s_net_bundle_nobm.py:324-325: add_n(REGULARIZATION_LOSSES): slim conv weights 1e-4*l2_loss(w) ([external]
    resnet_arg_scope; biases/BN/fully_connected carry none), fc_weights & fc_bias 2e-4*l2_loss (resnet.py:35-37)."""
    tot = 0.0
    for name, v in p.items():
        v64 = np.asarray(v, np.float64)
        if name.startswith('resnet_v2_50/') and name.endswith('/weights'):
            tot += cfg.weight_decay_conv * 0.5 * np.sum(v64 * v64)
        elif name in ('fc/fc_weights', 'fc/fc_bias'):
            tot += cfg.weight_decay_fc * 0.5 * np.sum(v64 * v64)
    return F(tot)


# --------------------------------------------------------------------------- optimiser
def exponential_decay_staircase(lr0, global_step, decay_steps, decay_rate):
    """[external] tf.train.exponential_decay(..., staircase=True) as called at train_bundle_nobm.py:155-158:
    float32 `lr0 * pow(decay_rate, floor(global_step / decay_steps))`."""
    p = F(np.floor(F(global_step) / F(decay_steps)))
    return F(F(lr0) * F(np.power(F(decay_rate), p)))


class AdamTF:
    """[external] tf.train.AdamOptimizer (TF 1.3) as used at train_bundle_nobm.py:159-160, defaults beta1 0.9,
    beta2 0.999, epsilon 1e-8.  Restated from the published ApplyAdam CPU kernel (training_ops.cc, Eigen expressions,
    one rounding per op, no FMA):
        alpha = lr * sqrt(1 - beta2_power) / (1 - beta1_power)
        m += (g - m) * (1 - beta1);  v += (g*g - v) * (1 - beta2);  var -= (m * alpha) / (sqrt(v) + epsilon)
    beta1_power / beta2_power are float32 variables initialised to beta1 / beta2 and multiplied by beta after every
    step (AdamOptimizer._finish).  dtype float32 reproduces TF's arithmetic; float64 is the exact-arithmetic shadow."""

    def __init__(self, n, beta1=0.9, beta2=0.999, epsilon=1e-8, dtype=np.float32):
        self.T = dtype
        self.beta1, self.beta2, self.eps = dtype(beta1), dtype(beta2), dtype(epsilon)
        self.m = np.zeros(n, dtype)
        self.v = np.zeros(n, dtype)
        self.b1p, self.b2p = dtype(beta1), dtype(beta2)

    def step(self, var, grad, lr):
        T = self.T
        var = np.asarray(var, T)
        g = np.asarray(grad, T)
        one = T(1)
        alpha = T(T(lr) * np.sqrt(one - self.b2p, dtype=T) / (one - self.b1p))
        self.m = (self.m + (g - self.m) * (one - self.beta1)).astype(T)
        self.v = (self.v + (g * g - self.v) * (one - self.beta2)).astype(T)
        var = (var - (self.m * alpha) / (np.sqrt(self.v, dtype=T) + self.eps)).astype(T)
        self.b1p = T(self.b1p * self.beta1)
        self.b2p = T(self.b2p * self.beta2)
        return var


# --------------------------------------------------------------------------- tower
def inference_stable_net(x_tensor, p, cfg: Config, y=None, matches=None, mask=None,
                         use_black_loss=1.0, use_theta_only=0.0, training=False):
    """s_net_bundle_nobm.py:266-385, one tower.  With y/matches/mask=None only the inference outputs are made
    (the contract of deploy_bundle.py:48-56,286)."""
    x_tensor = _f(x_tensor)
    cur = cfg.before_ch + cfg.before_ch if cfg.input_mask else cfg.before_ch
    x = x_tensor[..., cur:cur + 1]                                                   # :281
    theta, id_loss, id2_loss = get_resnet(x_tensor, p, cfg, training)
    pts1, pts2 = get_4_pts(theta, cfg)
    out, black, flow, Hs, _ = transformer(x, pts2, cfg, return_all=True)
    ret = {'theta': theta, 'pts1': pts1, 'pts2': pts2, 'output': out, 'black_pix': black,
           'x_map': flow[..., 0:1], 'y_map': flow[..., 1:2], 'Hs': Hs,
           'theta_loss': id_loss * F(cfg.theta_mul), 'grid_theta_loss': id2_loss * F(cfg.grid_theta_mul)}
    if y is None:
        return ret
    bp = get_black_pos(pts1, cfg)
    bp = bp * bp * F(use_black_loss)
    black_pos_loss = F(np.mean(bp, dtype=np.float64))
    regu = regu_loss(p, cfg)
    dist = get_distortion_loss(pts1, cfg)
    cons = get_consistency_loss(pts2, cfg)
    feat, warped = feature_loss(matches, mask, flow, cfg)
    il = img_loss(out, y, black, cfg)
    total = (id_loss * F(cfg.theta_mul) + id2_loss * F(cfg.grid_theta_mul) + (F(1) - F(use_theta_only)) * (
        il * F(cfg.img_mul) + regu * F(cfg.regu_mul) + black_pos_loss * F(cfg.black_mul)
        + dist * F(cfg.distortion_mul) + cons * F(cfg.consistency_mul) + feat * F(cfg.feature_mul)))
    ret.update({'error': np.abs(out - _f(y)), 'black_pos': bp, 'black_loss': black_pos_loss * F(cfg.black_mul),
                'distortion_loss': dist * F(cfg.distortion_mul), 'consistency_loss': cons * F(cfg.consistency_mul),
                'feature_loss': feat * F(cfg.feature_mul), 'img_loss': il * F(cfg.img_mul),
                'regu_loss': regu * F(cfg.regu_mul), 'total_loss': F(total), 'stable_warpped': warped})
    return ret


# --------------------------------------------------------------------------- deploy loop
class DeployRing:
    """deploy_bundle.py:204-232,259-274,291-295,319-332: history of stabilised frames + black masks,
    `before_ch = max(indices[1:])` deep (:34,:41), sampled at the dilated lags."""

    def __init__(self, first_frame, cfg: Config):
        self.cfg = cfg
        self.lags = [i for i in cfg.indices[1:] if i > 0]
        depth = max(self.lags)
        f0 = _f(first_frame).reshape(1, cfg.height, cfg.width, 1)
        self.frames = [f0.copy() for _ in range(depth)]                               # :216-217
        self.masks = [np.zeros((1, cfg.height, cfg.width, 1), F) for _ in range(depth)]   # :218

    def stack(self, cur):
        parts = [self.masks[-i] for i in self.lags] + [self.frames[-i] for i in self.lags]
        parts.append(_f(cur).reshape(1, self.cfg.height, self.cfg.width, 1))
        return np.concatenate(parts, axis=3)                                          # :259-274

    def push(self, frame, black):
        self.frames.append(_f(frame).reshape(1, self.cfg.height, self.cfg.width, 1))  # :322
        self.masks.append(_f(black).reshape(1, self.cfg.height, self.cfg.width, 1))   # :323
        self.frames.pop(0)
        self.masks.pop(0)


def deploy_step(ring: DeployRing, cur, p, cfg: Config, refine=1):
    """One iteration of the hot loop, deploy_bundle.py:259-296,319-332 (network + feedback only)."""
    in_x = ring.stack(cur)
    tmp = in_x.copy()
    for _ in range(refine):
        r = inference_stable_net(tmp, p, cfg)
        img = r['output'][0, :, :, 0]
        black = r['black_pix'][0]
        frame = img + black * F(-1)                                                   # :293
        tmp[..., -1] = frame[None]
    ring.push(frame, black)
    return r, frame


# --------------------------------------------------------------------------- colour remap
def cv_resize_linear_f32(src, dw, dh):
    """[external] cv2.resize(src, (dw, dh)) with the default INTER_LINEAR on a float32 single-channel image, restated
    from OpenCV's published algorithm (resizeGeneric_/HResizeLinear/VResizeLinear): half-pixel centres
    fx = (dx + 0.5) * scale - 0.5, source index clamped with the weight zeroed at the borders, horizontal pass then
    vertical pass, float32 arithmetic.  (No INTER_AREA shortcut: that exists only for an exact 2x shrink.)"""
    src = _f(src)
    sh, sw = src.shape
    sx_scale = float(sw) / dw
    sy_scale = float(sh) / dh

    def taps(n_dst, n_src, scale):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(F)
        s = np.floor(f).astype(np.int32)
        f = f - s.astype(F)
        lo = s < 0
        f[lo] = 0
        s[lo] = 0
        hi = s >= n_src - 1
        f[hi] = 0
        s[hi] = n_src - 1
        return s, np.minimum(s + 1, n_src - 1), (F(1.0) - f).astype(F), f.astype(F)

    x0, x1, ax0, ax1 = taps(dw, sw, sx_scale)
    y0, y1, by0, by1 = taps(dh, sh, sy_scale)
    rows = src[:, x0] * ax0[None, :] + src[:, x1] * ax1[None, :]                  # horizontal pass on every source row
    return rows[y0, :] * by0[:, None] + rows[y1, :] * by1[:, None]


_CV_INTER_BITS = 5
_CV_INTER_TAB_SIZE = 1 << _CV_INTER_BITS                 # 32 sub-pixel positions per axis
_CV_REMAP_COEF_BITS = 15
_CV_REMAP_COEF_SCALE = 1 << _CV_REMAP_COEF_BITS          # 32768
_cv_bilinear_tab_cache = None


def cv_bilinear_tab_i():
    """[external] OpenCV's fixed-point bilinear weight table BilinearTab_i[32*32][2][2] (imgproc/imgwarp.cpp,
    initInterTab2D(INTER_LINEAR, fixpt=true)), restated from the published source, literally -- including its two quirks:
    entry (fy, fx) holds saturate_cast<short>(wy[k1] * wx[k2] * 32768) for the four taps (k1 = row, k2 = column) with
    wy = (1 - fy/32, fy/32); every product is an exact multiple of 32 EXCEPT the weight 1.0 of entry (0, 0), which saturates to
    32767; entries whose sum is not 32768 are repaired by the bicubic-shaped loop `for k1, k2 in {ksize/2, ksize/2 + 1}`, which for
    ksize = 2 walks flat indices 3, 4, 5, 6 (its own last tap and the first three taps of the NEXT entry, still zero at that point
    of the one-time initialisation) and therefore adds the missing 1 to tap [1][1]: entry (0, 0) = {32767, 0, 0, 1}.
    -> int64 [1024, 4] (tap order [0][0], [0][1], [1][0], [1][1]; index = fy * 32 + fx)."""
    global _cv_bilinear_tab_cache
    if _cv_bilinear_tab_cache is not None:
        return _cv_bilinear_tab_cache
    n, ks = _CV_INTER_TAB_SIZE, 2
    scale = F(1.0) / F(n)
    tab1 = [(F(1.0) - F(i) * scale, F(i) * scale) for i in range(n)]          # interpolateLinear(i * scale)
    flat = np.zeros(n * n * ks * ks + 8, np.int64)                            # static storage: zero-initialised
    for i in range(n):
        for j in range(n):
            base = (i * n + j) * ks * ks
            isum = 0
            for k1 in range(ks):
                vy = tab1[i][k1]
                for k2 in range(ks):
                    v = F(vy * tab1[j][k2])
                    iv = int(np.clip(np.rint(np.float64(F(v * F(_CV_REMAP_COEF_SCALE)))), -32768, 32767))   # saturate_cast<short>
                    flat[base + k1 * ks + k2] = iv
                    isum += iv
            if isum != _CV_REMAP_COEF_SCALE:
                diff = isum - _CV_REMAP_COEF_SCALE
                k0 = ks // 2
                Mk = mk = (k0, k0)
                for k1 in range(k0, k0 + 2):
                    for k2 in range(k0, k0 + 2):
                        if flat[base + k1 * ks + k2] < flat[base + mk[0] * ks + mk[1]]:
                            mk = (k1, k2)
                        elif flat[base + k1 * ks + k2] > flat[base + Mk[0] * ks + Mk[1]]:
                            Mk = (k1, k2)
                if diff < 0:
                    flat[base + Mk[0] * ks + Mk[1]] -= diff
                else:
                    flat[base + mk[0] * ks + mk[1]] -= diff
    _cv_bilinear_tab_cache = flat[:n * n * ks * ks].reshape(n * n, ks * ks).copy()
    return _cv_bilinear_tab_cache


def cv_remap_linear_u8(img, map_x, map_y):
    """[external] cv2.remap(img_u8, map_x, map_y, INTER_LINEAR) with float32 maps, BORDER_CONSTANT 0 (deploy_bundle.py:144),
    restated from OpenCV's published algorithm (imgproc/imgwarp.cpp: RemapInvoker + remapBilinear<FixedPtCast<int, uchar, 15>,
    RemapVec_8u, short>): coordinates are quantised to 1/32 pixel, sx = cvRound(x * 32) (float32 product, half to even),
    integer part saturate_cast<short>(sx >> 5), table index (sy & 31) * 32 + (sx & 31); the four taps (out-of-frame taps read
    the border value 0) are blended with the 15-bit integer weights of cv_bilinear_tab_i() and the sum is rounded as
    (sum + 16384) >> 15, saturated to uint8.  Integer arithmetic throughout: bit-exact by construction."""
    img = np.asarray(img, np.uint8)
    H, W, C = img.shape
    # (cvRound of a float beyond the int range is INT_MIN on x86; such a coordinate is out of frame either way: clamp first)
    qx = np.clip(_f(map_x) * F(_CV_INTER_TAB_SIZE), F(-2.0e9), F(2.0e9))
    qy = np.clip(_f(map_y) * F(_CV_INTER_TAB_SIZE), F(-2.0e9), F(2.0e9))
    sx = np.rint(np.nan_to_num(qx, nan=-2.0e9)).astype(np.int64)
    sy = np.rint(np.nan_to_num(qy, nan=-2.0e9)).astype(np.int64)
    ix = np.clip(sx >> _CV_INTER_BITS, -32768, 32767)                          # saturate_cast<short>
    iy = np.clip(sy >> _CV_INTER_BITS, -32768, 32767)
    wtab = cv_bilinear_tab_i()[(sy & 31) * _CV_INTER_TAB_SIZE + (sx & 31)]       # [h, w, 4]

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        return np.where(ok[..., None], v, 0)

    acc = (tap(iy, ix) * wtab[..., 0:1] + tap(iy, ix + 1) * wtab[..., 1:2] + tap(iy + 1, ix) * wtab[..., 2:3]
           + tap(iy + 1, ix + 1) * wtab[..., 3:4])
    return np.clip((acc + (1 << (_CV_REMAP_COEF_BITS - 1))) >> _CV_REMAP_COEF_BITS, 0, 255).astype(np.uint8)


def warpRevBundle2(img, x_map, y_map, rate=4):
    """deploy_bundle.py:136-146: maps shrunk by `rate` and blown up again (seam smoothing), converted to pixel
    coordinates (m + 1) / 2 * size, bilinear remap of the colour frame.  img [H,W,3] uint8; maps [H,W] normalised."""
    H, W = x_map.shape
    xs = cv_resize_linear_f32(cv_resize_linear_f32(x_map, int(W / rate), int(H / rate)), W, H)
    ys = cv_resize_linear_f32(cv_resize_linear_f32(y_map, int(W / rate), int(H / rate)), W, H)
    xs = (xs + F(1)) / F(2) * F(W)
    ys = (ys + F(1)) / F(2) * F(H)
    return cv_remap_linear_u8(img, xs, ys), xs, ys


# --------------------------------------------------------------------------- crop search
def max_inscribed_rect(all_black, step=10):
    """deploy_bundle.py:344-366, literal restatement: largest black-free axis-aligned rectangle whose top-left corner
    lies on the `step` grid of the top-left quadrant; the first rectangle (i, j, hh, ww ascending) of strictly larger
    area wins.  all_black [H,W] integer counts.  Returns ([i, j, hh, ww], area); ([], 0) when no start pixel is free."""
    all_black = np.asarray(all_black, np.int64)
    H, W = all_black.shape
    S = np.zeros((H + 1, W + 1), np.int64)
    S[1:, 1:] = all_black.cumsum(0).cumsum(1)
    max_s, ans = 0, []
    for i in range(0, int(math.floor(H * 0.5)), step):
        for j in range(0, int(math.floor(W * 0.5)), step):
            if all_black[i, j] > 0:
                continue
            for hh in range(i, H):
                # first column ww >= j at which rows i..hh contain black (the inner `break`), vectorised
                row = S[hh + 1, j + 1:] - S[hh + 1, j] - S[i, j + 1:] + S[i, j]
                bad = np.nonzero(row > 0)[0]
                n_free = int(bad[0]) if bad.size else W - j
                if n_free == 0:
                    continue
                s = (hh - i + 1) * n_free
                if s > max_s:                   # within one hh the area grows with ww: the last free column is the max
                    max_s, ans = s, [i, j, hh, j + n_free - 1]
    return ans, max_s


# --------------------------------------------------------------------------- training sample assembly
# get_data_mini_after.py:7-147,229-253.  The reference draws its random numbers inside the TF graph (Philox streams that
# cannot be reproduced without TF); here every random quantity is an INPUT (crop offsets, flip, the contrast factor and
# brightness delta, the mask homographies), so the arithmetic after the draw is what is restated and compared.
RANDOM_CROP_RATE = 0.9                      # configs/v2_93.py:23
RAND_H_MAX = np.array([[1.1, 0.1, 0.5], [0.1, 1.1, 0.5], [0.1, 0.1, 1]], np.float32)      # configs/v2_93.py:37
RAND_H_MIN = np.array([[0.9, -0.1, -0.5], [-0.1, 0.9, -0.5], [-0.1, -0.1, 1]], np.float32)  # configs/v2_93.py:38


def tf_resize_bilinear(img, oh, ow):
    """[external] tf.image.resize_images(BILINEAR) of TF 1.3 = ResizeBilinear(align_corners=False): scale = in/out (float32),
    in = out_index * scale, top = floor, bottom = min(ceil, in-1), lerp = in - top;
    out = top_row + (bottom_row - top_row) * y_lerp with row values tl + (tr - tl) * x_lerp.  img [H,W] float32."""
    img = np.asarray(img, F)
    ih, iw = img.shape
    hs, ws = F(ih) / F(oh), F(iw) / F(ow)
    iy = (np.arange(oh, dtype=F) * hs).astype(F)
    ix = (np.arange(ow, dtype=F) * ws).astype(F)
    y0 = np.floor(iy).astype(np.int64); y1 = np.minimum(np.ceil(iy).astype(np.int64), ih - 1); ly = (iy - y0.astype(F)).astype(F)
    x0 = np.floor(ix).astype(np.int64); x1 = np.minimum(np.ceil(ix).astype(np.int64), iw - 1); lx = (ix - x0.astype(F)).astype(F)
    tl, tr = img[y0][:, x0], img[y0][:, x1]
    bl, br = img[y1][:, x0], img[y1][:, x1]
    top = (tl + ((tr - tl).astype(F) * lx[None, :]).astype(F)).astype(F)
    bot = (bl + ((br - bl).astype(F) * lx[None, :]).astype(F)).astype(F)
    return (top + ((bot - top).astype(F) * ly[:, None]).astype(F)).astype(F)


def aug_resized_hw(H, W, rate=RANDOM_CROP_RATE):
    return int(H / rate), int(W / rate)         # get_data_mini_after.py:8-9


def warp_img(image, para, contrast, brightness, rate=RANDOM_CROP_RATE):
    """get_data_mini_after.py:14-31 for one [H,W] channel: resize up, crop at (para h, w), flip, tf.image contrast
    ((x - mean)*factor + mean, per-channel mean over H,W) and brightness (x + delta), clip to [-0.5, 0.5]."""
    H, W = image.shape
    h, w = aug_resized_hw(H, W, rate)
    big = tf_resize_bilinear(image, h, w)
    img = big[para["h"]:para["h"] + H, para["w"]:para["w"] + W]
    if para["flip"]:
        img = img[:, ::-1]
    mean = F(np.asarray(img, np.float64).mean())
    img = (((img - mean).astype(F) * F(contrast)).astype(F) + mean).astype(F)
    img = (img + F(brightness)).astype(F)
    return np.clip(img, F(-0.5), F(0.5)).astype(F)


def warp_flow(flow, para, rate=RANDOM_CROP_RATE):
    """get_data_mini_after.py:33-51.  flow [H,W,2] in normalised coordinates."""
    H, W = flow.shape[:2]
    h, w = aug_resized_hw(H, W, rate)
    fx = tf_resize_bilinear(flow[..., 0], h, w)[para["h"]:para["h"] + H, para["w"]:para["w"] + W]
    fy = tf_resize_bilinear(flow[..., 1], h, w)[para["h"]:para["h"] + H, para["w"]:para["w"] + W]
    ox = F(1) - (F(para["w"]) / F(w)) * F(2)
    oy = F(1) - (F(para["h"]) / F(h)) * F(2)
    fx = (((fx + ox).astype(F) / F(H / float(h))).astype(F) - F(1)).astype(F)     # (sic) x is divided by height/h
    fy = (((fy + oy).astype(F) / F(W / float(w))).astype(F) - F(1)).astype(F)
    if para["flip"]:
        fy = fy[:, ::-1]
        fx = ((fx[:, ::-1] * F(-1)).astype(F) - F(1.0 / W)).astype(F)
    return np.stack([fx, fy], axis=2).astype(F)


def warp_point(points, mask, para, H, W, rate=RANDOM_CROP_RATE):
    """get_data_mini_after.py:53-70.  points [M,4] = (x1,y1,x2,y2) normalised; mask [M] bool."""
    h, w = aug_resized_hw(H, W, rate)
    p = np.asarray(points, F)
    ox = F(1) - (F(para["w"]) / F(w)) * F(2)
    oy = F(1) - (F(para["h"]) / F(h)) * F(2)
    px = (((p[:, [0, 2]] + ox).astype(F) / F(H / float(h))).astype(F) - F(1)).astype(F)
    py = (((p[:, [1, 3]] + oy).astype(F) / F(W / float(w))).astype(F) - F(1)).astype(F)
    if para["flip"]:
        px = ((px * F(-1)).astype(F) - F(1.0 / W)).astype(F)
    out = np.stack([px[:, 0], py[:, 0], px[:, 1], py[:, 1]], axis=1).astype(F)
    ok = np.logical_and(np.all(np.logical_and(out >= -1, out <= 1), axis=1), np.asarray(mask, bool))
    return out, ok


def rand_mask_from_H(Hm, H, W):
    """get_data_mini_after.py:93-108: black where H*grid leaves [-1,1] (strict), grid = (linspace x, linspace y, 1)."""
    Hm = np.asarray(Hm, F)
    gx = linspace_tf(-1.0, 1.0, W)[None, :].repeat(H, 0)
    gy = linspace_tf(-1.0, 1.0, H)[:, None].repeat(W, 1)

    def row(r):      # tf.matmul [3,3]x[3,HW]: one dot product of length 3 per element, accumulated in order
        return ((Hm[r, 0] * gx).astype(F) + (Hm[r, 1] * gy).astype(F) + Hm[r, 2]).astype(F)
    xs, ys, zs = row(0), row(1), row(2)
    x = (xs / zs).astype(F); y = (ys / zs).astype(F)
    return ((F(-1) > x) | (x > F(1)) | (F(-1) > y) | (y > F(1))).astype(F)


def add_mask(pics, Hs, input_mask=True):
    """get_data_mini_after.py:128-147.  pics [H,W,before_ch]; Hs [before_ch,3,3] (rand_H_change_rate = 1: every channel a
    fresh homography).  Returns [H,W,2*before_ch] = masks then masked frames (masked pixels = -1)."""
    H, W, C = pics.shape
    masks = np.stack([rand_mask_from_H(Hs[i], H, W) for i in range(C)], axis=2)
    ans = ((pics * (F(1) - masks)).astype(F) + (masks * F(-1)).astype(F)).astype(F)
    return np.concatenate([masks, ans], axis=2) if input_mask else ans


def assemble_pair(stable, unstable, flow, matches1, n1, matches2, n2, para, contrast, brightness, Hs1, Hs2, cfg: Config):
    """get_data_mini_after.py:229-253.  stable [H,W,2*(before_ch+1)] (y1, 6 history, y2, 6 history), unstable [H,W,2]."""
    bc = cfg.before_ch
    H, W = stable.shape[:2]
    st = np.stack([warp_img(stable[..., i], para, contrast, brightness) for i in range(stable.shape[2])], axis=2)
    un = np.stack([warp_img(unstable[..., i], para, contrast, brightness) for i in range(unstable.shape[2])], axis=2)
    x1 = np.concatenate([add_mask(st[..., 1:1 + bc], Hs1, cfg.input_mask), un[..., 0:1]], axis=2)
    y1 = st[..., 0:1]
    x2 = np.concatenate([add_mask(st[..., bc + 2:2 * bc + 2], Hs2, cfg.input_mask), un[..., 1:2]], axis=2)
    y2 = st[..., bc + 1:bc + 2]
    m1 = np.arange(cfg.max_matches) < n1
    m2 = np.arange(cfg.max_matches) < n2
    fm1, mk1 = warp_point(matches1, m1, para, H, W)
    fm2, mk2 = warp_point(matches2, m2, para, H, W)
    return x1, y1, x2, y2, warp_flow(flow, para), fm1, mk1, fm2, mk2
