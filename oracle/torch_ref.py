"""Gradient oracle: the training objective of the reference restated in torch (CPU, float64, autograd).

TEST INFRASTRUCTURE ONLY (same rules as stabnet_oracle.py; parity unpinned for the same reason).  The NumPy oracle
defines the forward values; this file exists because the reference differentiates its graph with TF autodiff
(`opt.minimize(total_loss)`, train_bundle_nobm.py:160) and the HIP backward kernels need an independent gradient to
be checked against.  tests/test_oracle_cpu.py checks this file's forward against stabnet_oracle.py.

Autodiff conventions reproduced (they decide the gradient): floor / int casts / comparisons carry no gradient, so
sampler corners, `black_pix`, `warp_pts` indices and the z-sign are constants; clip_by_value / minimum / maximum pass
gradient only where not saturated; gather back-propagates as scatter-add.

DECISIONS.  The objective is piecewise smooth: ReLU signs, the max-pool arg-max, `black_pix`, the sampler's floor corners.
A float32 forward and this float64 one can take different sides of such a decision when the deciding value sits within
float32 rounding of its threshold; the two then differentiate DIFFERENT smooth pieces and their gradients differ by far more
than rounding.  `decisions=` (train_objective) lets a test hand in the decisions the float32 forward under test really took
(per tower: 'relu' {BN prefix | 'fc1'..'fc3' -> bool array}, 'pool_argmax' uint8 [N,Ho,Wo,C] = dy*3+dx, 'black' [N,H,W],
'corners' (x0,y0,x1,y1)), so that both sides differentiate the same piece; `record=` collects this file's own decisions in the
same format, so a test can name the ones that flipped.  Decisions that depend on INPUT data only (feature-loss pixel indices,
the corners of `interpolate(., flow)`) are taken in float32 here, as the reference takes them.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as Fnn

from . import stabnet_oracle as O

DT = torch.float64


def t(x, requires_grad=False):
    return torch.tensor(np.asarray(x), dtype=DT, requires_grad=requires_grad)


def get_4_pts(theta, cfg):
    """s_net_bundle_nobm.py:29-71 -> pts1 [N,gh,gw,8], pts2 [N,gh+1,gw+1,2]."""
    N = theta.shape[0]
    gh, gw = cfg.grid_h, cfg.grid_w
    lim = 1.0 / float(np.float32(cfg.do_crop_rate))
    base = torch.tensor([[[j * (2.0 / gw) - 1, i * (2.0 / gh) - 1] for j in range(gw + 1)] for i in range(gh + 1)], dtype=DT)
    pts2 = base[None] + theta.reshape(N, gh + 1, gw + 1, 2)
    pts2 = torch.minimum(torch.maximum(pts2, torch.tensor(-lim, dtype=DT)), torch.tensor(lim, dtype=DT))
    tl, tr, bl, br = pts2[:, :-1, :-1], pts2[:, :-1, 1:], pts2[:, 1:, :-1], pts2[:, 1:, 1:]
    pts1 = torch.stack([tl, tr, bl, br], dim=-1).reshape(N, gh, gw, 8)
    return pts1, pts2


def get_Hs(pts2, cfg):
    """spatial_transformer3.py:144-198 -> [N,gh,gw,9]."""
    N = pts2.shape[0]
    gh, gw = cfg.grid_h, cfg.grid_w
    h, w = 2.0 / gh, 2.0 / gw
    out = []
    ridge = torch.eye(8, dtype=DT) * float(np.float32(1e-4))
    for i in range(gh):
        for j in range(gw):
            hh, ww = i * h - 1, j * w - 1
            sx = torch.tensor([ww, ww + w, ww, ww + w], dtype=DT)
            sy = torch.tensor([hh, hh, hh + h, hh + h], dtype=DT)
            tar = torch.stack([pts2[:, i, j], pts2[:, i, j + 1], pts2[:, i + 1, j], pts2[:, i + 1, j + 1]], dim=1)  # [N,4,2]
            u, v = tar[:, :, 0], tar[:, :, 1]
            A = torch.zeros(N, 8, 8, dtype=DT)
            one = torch.ones(N, 4, dtype=DT)
            A[:, :4, 0] = sx; A[:, :4, 1] = sy; A[:, :4, 2] = one
            A[:, 4:, 3] = sx; A[:, 4:, 4] = sy; A[:, 4:, 5] = one
            A = A.clone()
            A6u, A7u = -sx[None] * u, -sy[None] * u
            A6v, A7v = -sx[None] * v, -sy[None] * v
            A = torch.cat([A[:, :, :6], torch.cat([A6u, A6v], dim=1)[:, :, None], torch.cat([A7u, A7v], dim=1)[:, :, None]], dim=2)
            b = torch.cat([u, v], dim=1)[:, :, None]
            hvec = torch.linalg.solve(A + ridge[None], b)[:, :, 0]
            out.append(torch.cat([hvec, torch.ones(N, 1, dtype=DT)], dim=1))
    return torch.stack(out, dim=1).reshape(N, gh, gw, 9)


def _sample(im, x, y, corners=None):
    """spatial_transformer3.py:62-123; im [N,H,W,C]; x,y [N,H,W] -> [N,H,W,C].
    corners = (x0, y0, x1, y1) clipped int arrays: the (non-differentiable) floor decisions taken in float32 by the
    forward pass; given, they replace this function's float64 floor so that both sides differentiate the same graph."""
    N, H, W, C = im.shape
    xp = (x + 1.0) * W / 2.0
    yp = (y + 1.0) * H / 2.0
    with torch.no_grad():
        if corners is not None:
            x0, y0, x1, y1 = [torch.as_tensor(np.asarray(c).reshape(N, H, W).astype(np.int64)) for c in corners]
        else:
            x0 = torch.floor(xp).clamp(-2 ** 31, 2 ** 31 - 1).long()
            y0 = torch.floor(yp).clamp(-2 ** 31, 2 ** 31 - 1).long()
            x1, y1 = x0 + 1, y0 + 1
            x0, x1 = x0.clamp(0, W - 1), x1.clamp(0, W - 1)
            y0, y1 = y0.clamp(0, H - 1), y1.clamp(0, H - 1)
        base = (torch.arange(N) * H * W)[:, None, None]
        ia, ib, ic, idd = base + y0 * W + x0, base + y1 * W + x0, base + y0 * W + x1, base + y1 * W + x1
    flat = im.reshape(-1, C)
    Ia, Ib, Ic, Id = flat[ia.reshape(-1)], flat[ib.reshape(-1)], flat[ic.reshape(-1)], flat[idd.reshape(-1)]
    x0f, x1f, y0f, y1f = x0.to(DT), x1.to(DT), y0.to(DT), y1.to(DT)
    wa = ((x1f - xp) * (y1f - yp)).reshape(-1, 1)
    wb = ((x1f - xp) * (yp - y0f)).reshape(-1, 1)
    wc = ((xp - x0f) * (y1f - yp)).reshape(-1, 1)
    wd = ((xp - x0f) * (yp - y0f)).reshape(-1, 1)
    return (wa * Ia + wb * Ib + wc * Ic + wd * Id).reshape(N, H, W, C)


def corners_f32(x, y, H, W):
    """The sampler's clipped floor corners decided in float32 with the reference's op order (spatial_transformer3.py:81-93);
    x, y: float arrays [N,H,W] (normalised coordinates)."""
    x = np.asarray(x, np.float32)
    y = np.asarray(y, np.float32)
    xp = (x + np.float32(1.0)) * np.float32(W) / np.float32(2.0)
    yp = (y + np.float32(1.0)) * np.float32(H) / np.float32(2.0)
    x0 = np.clip(np.floor(xp), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int64)
    y0 = np.clip(np.floor(yp), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int64)
    return (np.clip(x0, 0, W - 1), np.clip(y0, 0, H - 1), np.clip(x0 + 1, 0, W - 1), np.clip(y0 + 1, 0, H - 1))


def _own_corners(x, y, H, W):
    xp = (x.detach() + 1.0) * W / 2.0
    yp = (y.detach() + 1.0) * H / 2.0
    x0 = torch.floor(xp).clamp(-2 ** 31, 2 ** 31 - 1).long()
    y0 = torch.floor(yp).clamp(-2 ** 31, 2 ** 31 - 1).long()
    return tuple(a.numpy() for a in (x0.clamp(0, W - 1), y0.clamp(0, H - 1), (x0 + 1).clamp(0, W - 1), (y0 + 1).clamp(0, H - 1)))


def transformer(U, pts2, cfg, corners=None, black=None, record=None):
    """spatial_transformer3.py:218-301 -> (out [N,H,W,C], black [N,H,W], maps [N,H,W,2]).  corners / black: decisions of the
    forward under test (module docstring); record: dict that receives this evaluation's own."""
    N, H, W, C = U.shape
    gh, gw = cfg.grid_h, cfg.grid_w
    Hs = get_Hs(pts2, cfg)
    xs = torch.tensor(O.linspace_tf(-1.0, 1.0, W).astype(np.float64))
    ys = torch.tensor(O.linspace_tf(-1.0, 1.0, H).astype(np.float64))
    rows, cols = O.cell_bounds(H, W, gh, gw)
    xrows = []
    yrows = []
    for i, (sh, eh) in enumerate(rows):
        xr, yr = [], []
        for j, (sw, ew) in enumerate(cols):
            h = Hs[:, i, j, :][:, :, None, None]
            gx = xs[sw:ew + 1][None, None, :]
            gy = ys[sh:eh + 1][None, :, None]
            tx = h[:, 0] * gx + h[:, 1] * gy + h[:, 2]
            ty = h[:, 3] * gx + h[:, 4] * gy + h[:, 5]
            tz = h[:, 6] * gx + h[:, 7] * gy + h[:, 8]
            sign = torch.where(tz >= 0, 1.0, -1.0).detach()
            tz = tz + sign * float(np.float32(1e-8))
            xr.append(tx / tz)
            yr.append(ty / tz)
        xrows.append(torch.cat(xr, dim=2))
        yrows.append(torch.cat(yr, dim=2))
    x_map = torch.cat(xrows, dim=1)
    y_map = torch.cat(yrows, dim=1)
    own_black = ((x_map < -1) | (x_map > 1) | (y_map < -1) | (y_map > 1)).detach()
    if record is not None:
        record['black'] = own_black.numpy().copy()
        record['corners'] = _own_corners(x_map, y_map, H, W)
    black = (torch.as_tensor(np.asarray(black).reshape(N, H, W) != 0) if black is not None else own_black).to(DT)
    out = _sample(U, x_map, y_map, corners)
    return out, black, torch.stack([x_map, y_map], dim=3), Hs


def interpolate(im, x, y, corners=None):
    """spatial_transformer.py:200-281.  The maps are input data (the optical flow): their floor corners are float32 decisions of
    the reference, taken here in float32 too unless given."""
    N, H, W, C = im.shape
    x, y = x.reshape(N, H, W), y.reshape(N, H, W)
    if corners is None:
        corners = corners_f32(x.detach().numpy(), y.detach().numpy(), H, W)
    return _sample(im, x, y, corners)


# ---- losses ------------------------------------------------------------------------------------------------
def get_black_pos(pts1, cfg):
    lim = 1.0 / float(np.float32(cfg.do_crop_rate))
    z = torch.zeros((), dtype=DT)
    e = torch.where(pts1 > lim, pts1 - lim, z) + torch.where(-lim > pts1, -lim - pts1, z)
    return e.reshape(pts1.shape[0], -1)


def get_distortion_loss(pts1, cfg):
    pts = pts1.reshape(-1, 2, 4)
    p = [pts[:, :, k] for k in range(4)]
    h, w = 2.0 / cfg.grid_h, 2.0 / cfg.grid_w

    def calc(p0, p1, p2, clock, hw):
        k = h / w if hw == 0 else w / h
        R = torch.tensor([[0, -k], [k, 0]] if not clock else [[0, k], [-k, 0]], dtype=DT)
        loss = torch.abs((p1 - p0) @ R.T - (p2 - p1))
        return loss * loss
    p0, p1, p2, p3 = p
    loss = calc(p0, p1, p3, 0, 0) + calc(p1, p3, p2, 0, 1) + calc(p3, p2, p0, 0, 0) + calc(p2, p0, p1, 0, 1)
    loss = loss + calc(p1, p0, p2, 1, 0) + calc(p0, p2, p3, 1, 1) + calc(p2, p3, p1, 1, 0) + calc(p3, p1, p0, 1, 1)
    return loss.mean() / 8


def get_consistency_loss(pts2, cfg):
    gh, gw = cfg.grid_h, cfg.grid_w
    p = pts2
    errs = []
    for i in range(gh + 1):
        for j in range(gw + 1):
            if i > 1:
                errs.append(torch.abs(2 * p[:, i - 1, j] - p[:, i, j] - p[:, i - 2, j]))
            if j > 1:
                errs.append(torch.abs(2 * p[:, i, j - 1] - p[:, i, j] - p[:, i, j - 2]))
            if i < gh - 1:
                errs.append(torch.abs(2 * p[:, i + 1, j] - p[:, i, j] - p[:, i + 2, j]))
            if j < gw - 1:
                errs.append(torch.abs(2 * p[:, i, j + 1] - p[:, i, j] - p[:, i, j + 2]))
    e = torch.stack(errs, dim=2)
    return (e * e).mean()


def feature_loss(matches, mask, flow, cfg):
    N, H, W, _ = flow.shape
    stable, unstable = matches[:, :, :2], matches[:, :, 2:]
    with torch.no_grad():
        x = torch.clamp((stable[:, :, 0] + 1) / 2 * W, 0, W - 1)
        y = torch.clamp((stable[:, :, 1] + 1) / 2 * H, 0, H - 1)
        # indices are float32 decisions in the reference (tf.round on float32): decide them in float32
        xi = torch.tensor(np.rint(np.clip((stable[:, :, 0].numpy().astype(np.float32) + np.float32(1)) / np.float32(2) * np.float32(W), 0, W - 1)).astype(np.int64))
        yi = torch.tensor(np.rint(np.clip((stable[:, :, 1].numpy().astype(np.float32) + np.float32(1)) / np.float32(2) * np.float32(H), 0, H - 1)).astype(np.int64))
    warped = torch.stack([flow[n].reshape(-1, 2)[xi[n] + yi[n] * W] for n in range(N)], dim=0)
    before = torch.abs(warped - unstable).sum(dim=2)
    after = (before * mask).sum(dim=1) / torch.clamp(mask.sum(dim=1), min=1.0)
    return after.mean()


def img_loss(out, y, black, cfg):
    N, H, W, _ = out.shape
    keep = (1 - black).reshape(N, H, W, 1)
    err = (out - y) * keep
    return ((err * err).sum(dim=(1, 2, 3)) / (keep.sum(dim=(1, 2, 3)) + 1e-8)).sum() / cfg.batch_size


def temporal_loss(out1, black1, out2, black2, flow, cfg, use_temp_loss=1.0):
    N, H, W, _ = out1.shape
    fx, fy = flow[..., 0], flow[..., 1]
    o2 = interpolate(out2, fx, fy)
    nb2 = interpolate((1 - black2).reshape(N, H, W, 1), fx, fy)
    noblack = (1 - black1).reshape(N, H, W, 1) * nb2
    err = (out1 - o2) * noblack
    return ((err * err).sum(dim=(1, 2, 3)) / (noblack.sum(dim=(1, 2, 3)) + 1e-8)).sum() / cfg.batch_size * use_temp_loss


# ---- backbone ------------------------------------------------------------------------------------------------
def _conv(x, w, stride=1, pads=(0, 0, 0, 0), bias=None):
    """NHWC x, HWIO w; pads = (top, bottom, left, right)."""
    xc = x.permute(0, 3, 1, 2)
    if any(pads):
        xc = Fnn.pad(xc, (pads[2], pads[3], pads[0], pads[1]))
    y = Fnn.conv2d(xc, w.permute(3, 2, 0, 1), bias, stride=stride)
    return y.permute(0, 2, 3, 1)


def _conv_same(x, w, stride, bias=None):
    k = w.shape[0]
    if stride == 1:
        ph = O._same_pads(x.shape[1], k, 1)
        pw = O._same_pads(x.shape[2], k, 1)
        return _conv(x, w, 1, (ph[0], ph[1], pw[0], pw[1]), bias)
    tot = k - 1
    beg = tot // 2
    end = tot - beg
    return _conv(x, w, stride, (beg, end, beg, end), bias)


def _bn(x, p, prefix, cfg, training, batch_stats=None):
    gamma, beta = p[prefix + '/gamma'], p[prefix + '/beta']
    if training:
        mean = x.mean(dim=(0, 1, 2))
        var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
        if batch_stats is not None:
            batch_stats[prefix] = (mean.detach().numpy().copy(), var.detach().numpy().copy())
    else:
        mean, var = p[prefix + '/moving_mean'], p[prefix + '/moving_variance']
    inv = torch.rsqrt(var + float(np.float32(cfg.bn_eps))) * gamma
    return x * inv + (beta - mean * inv)


def _relu(y, key, dec, rec):
    """relu(y).  dec['relu'][key] (bool, y's shape): the sign decisions of the forward under test replace this evaluation's own
    (y * mask has relu's derivative wherever the two agree); rec['relu'][key] receives this evaluation's own."""
    if rec is not None:
        rec.setdefault('relu', {})[key] = (y.detach() > 0).numpy()
    if dec is not None and key in dec.get('relu', {}):
        return y * torch.as_tensor(np.asarray(dec['relu'][key]).reshape(tuple(y.shape)) != 0).to(DT)
    return torch.relu(y)


def _max_pool_3x3s2(net, dec, rec):
    """slim max_pool2d(3, stride 2, 'SAME') on NHWC; dec['pool_argmax'] (uint8 [N,Ho,Wo,C], dy*3+dx in the window, first maximum
    in scan order) replaces the arg-max of this evaluation."""
    ph = O._same_pads(net.shape[1], 3, 2)
    pw = O._same_pads(net.shape[2], 3, 2)
    nc = Fnn.pad(net.permute(0, 3, 1, 2), (pw[0], pw[1], ph[0], ph[1]), value=-math.inf)
    pooled, idx = Fnn.max_pool2d(nc, 3, 2, return_indices=True)
    N, C, Ho, Wo = pooled.shape
    Wp = nc.shape[3]
    oy = torch.arange(Ho)[None, None, :, None]
    ox = torch.arange(Wo)[None, None, None, :]
    if rec is not None:
        iy, ix = idx // Wp, idx % Wp
        rec['pool_argmax'] = ((iy - 2 * oy) * 3 + (ix - 2 * ox)).permute(0, 2, 3, 1).numpy().astype(np.uint8)
    if dec is not None and 'pool_argmax' in dec:
        am = torch.as_tensor(np.asarray(dec['pool_argmax']).reshape(N, Ho, Wo, C).astype(np.int64)).permute(0, 3, 1, 2)
        # the kernel's window starts at (2 oy - pad_top, 2 ox - pad_left): in the padded image that is (2 oy, 2 ox)
        flat = (2 * oy + am // 3) * Wp + (2 * ox + am % 3)
        pooled = nc.reshape(N, C, -1).gather(2, flat.reshape(N, C, -1)).reshape(N, C, Ho, Wo)
    return pooled.permute(0, 2, 3, 1)


def resnet_v2_50(x, p, cfg, training, batch_stats=None, dec=None, rec=None):
    R = 'resnet_v2_50/'
    net = _conv_same(x, p[R + 'conv1/weights'], 2, p[R + 'conv1/biases'])
    net = _max_pool_3x3s2(net, dec, rec)
    for (bname, depth, dbn, units, bstride) in O.RESNET_V2_50_BLOCKS:
        for u in range(1, units + 1):
            stride = bstride if u == units else 1
            S = R + '%s/unit_%d/bottleneck_v2/' % (bname, u)
            depth_in = net.shape[3]
            preact = _relu(_bn(net, p, S + 'preact', cfg, training, batch_stats), S + 'preact', dec, rec)
            if depth == depth_in:
                shortcut = net if stride == 1 else net[:, ::stride, ::stride, :]
            else:
                shortcut = _conv(preact, p[S + 'shortcut/weights'], stride, bias=p[S + 'shortcut/biases'])
            r = _conv(preact, p[S + 'conv1/weights'], 1)
            r = _relu(_bn(r, p, S + 'conv1/BatchNorm', cfg, training, batch_stats), S + 'conv1/BatchNorm', dec, rec)
            r = _conv_same(r, p[S + 'conv2/weights'], stride)
            r = _relu(_bn(r, p, S + 'conv2/BatchNorm', cfg, training, batch_stats), S + 'conv2/BatchNorm', dec, rec)
            r = _conv(r, p[S + 'conv3/weights'], 1, bias=p[S + 'conv3/biases'])
            net = shortcut + r
    return _relu(_bn(net, p, R + 'postnorm', cfg, training, batch_stats), R + 'postnorm', dec, rec)


def get_resnet(x_tensor, p, cfg, training, batch_stats=None, dec=None, rec=None):
    feat = resnet_v2_50(x_tensor, p, cfg, training, batch_stats, dec, rec)
    g = feat.mean(dim=(1, 2))
    for k in (1, 2, 3):
        g = _relu(g @ p['fc/fc/fc_%d/weights' % k] + p['fc/fc/fc_%d/biases' % k], 'fc%d' % k, dec, rec)
    theta = g @ p['fc/fc_weights'] + p['fc/fc_bias']
    id2 = theta.abs().mean() * cfg.id_mul
    return theta, id2, id2


def regu_loss(p, cfg):
    tot = torch.zeros((), dtype=DT)
    for name, v in p.items():
        if name.startswith('resnet_v2_50/') and name.endswith('/weights'):
            tot = tot + cfg.weight_decay_conv * 0.5 * (v * v).sum()
        elif name in ('fc/fc_weights', 'fc/fc_bias'):
            tot = tot + cfg.weight_decay_fc * 0.5 * (v * v).sum()
    return tot


def tower_losses(theta, x_cur, y, matches, mask, cfg, use_black_loss=1.0, dec=None, rec=None):
    """Everything of inference_stable_net downstream of theta (s_net_bundle_nobm.py:304-352)."""
    pts1, pts2 = get_4_pts(theta, cfg)
    out, black, flow, Hs = transformer(x_cur, pts2, cfg, corners=(dec or {}).get('corners'), black=(dec or {}).get('black'),
                                       record=rec)
    bp = get_black_pos(pts1, cfg)
    black_pos_loss = (bp * bp * use_black_loss).mean()
    return {'pts1': pts1, 'pts2': pts2, 'output': out, 'black_pix': black, 'flow': flow, 'Hs': Hs,
            'black_pos_loss': black_pos_loss, 'distortion': get_distortion_loss(pts1, cfg),
            'consistency': get_consistency_loss(pts2, cfg), 'feature': feature_loss(matches, mask, flow, cfg),
            'img': img_loss(out, y, black, cfg)}


def tower_total(id_loss, id2_loss, L, regu, cfg, use_theta_only=0.0):
    """s_net_bundle_nobm.py:355-359."""
    return id_loss * cfg.theta_mul + id2_loss * cfg.grid_theta_mul + (1 - use_theta_only) * (
        L['img'] * cfg.img_mul + regu * cfg.regu_mul + L['black_pos_loss'] * cfg.black_mul
        + L['distortion'] * cfg.distortion_mul + L['consistency'] * cfg.consistency_mul + L['feature'] * cfg.feature_mul)


def train_objective(p, batch, cfg, use_temp_loss=1.0, use_black_loss=1.0, use_theta_only=0.0, training=True,
                    batch_stats=None, decisions=None, record=None):
    """train_bundle_nobm.py:107-142: two towers sharing weights + temporal loss.  p: name -> torch tensor (TF layout).
    decisions / record: {'1': {...}, '2': {...}} per tower, see the module docstring.  Returns (total, parts dict)."""
    parts = {}
    regu = regu_loss(p, cfg)
    towers = []
    for k in ('1', '2'):
        x = t(batch['x' + k])
        cur = 2 * cfg.before_ch if cfg.input_mask else cfg.before_ch
        bs = {} if batch_stats is not None else None
        dec = decisions.get(k) if decisions is not None else None
        rec = record.setdefault(k, {}) if record is not None else None
        theta, id_loss, id2_loss = get_resnet(x, p, cfg, training, bs, dec, rec)
        if batch_stats is not None:
            batch_stats[k] = bs
        L = tower_losses(theta, x[..., cur:cur + 1], t(batch['y' + k]), t(batch['matches' + k]), t(batch['mask' + k]),
                         cfg, use_black_loss, dec, rec)
        L['theta'] = theta
        L['total'] = tower_total(id_loss, id2_loss, L, regu, cfg, use_theta_only)
        towers.append(L)
    temp = temporal_loss(towers[0]['output'], towers[0]['black_pix'], towers[1]['output'], towers[1]['black_pix'],
                         t(batch['flow']), cfg, use_temp_loss)
    total = towers[0]['total'] + towers[1]['total'] + temp * cfg.temp_mul
    parts.update({'tower1': towers[0], 'tower2': towers[1], 'temp_loss': temp, 'regu_loss': regu, 'total': total})
    return total, parts
