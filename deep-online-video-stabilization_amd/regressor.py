"""Host side of the regressor: owns the net plan handle and the caller-side buffers (flat parameter buffer, folded
BN, workspace) and mirrors get_resnet (s_net_bundle_nobm.py:250-264).  Torch only allocates and holds pointers."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from ._tensor import dev_f32, empty, ptr, stream_ptr
from .config import Config, v2_93

KIND_CONV_W, KIND_BIAS, KIND_GAMMA, KIND_BETA, KIND_MEAN, KIND_VAR, KIND_FC_W, KIND_FC_B = range(8)


class NetPlan:
    """RAII wrapper of a stabnet_net handle + its parameter table."""

    def __init__(self, N: int, H: int, W: int, cfg: Config = v2_93, keep_activations: bool = False):
        self.N, self.H, self.W, self.cfg = N, H, W, cfg
        L = _lib.lib()
        self._h = ctypes.c_void_p()
        _lib.call("stabnet_net_create", ctypes.byref(self._h), N, H, W, cfg.in_ch, cfg.n_theta, int(keep_activations))
        self.n_floats = L.stabnet_net_param_floats(self._h)
        self.n_trainable = L.stabnet_net_trainable_floats(self._h)
        self.bn_channels = L.stabnet_net_bn_channels(self._h)
        self.workspace_bytes = L.stabnet_net_workspace_bytes(self._h)
        self.flops = L.stabnet_net_flops(self._h)
        self.num_launches = L.stabnet_net_num_launches(self._h)
        self.table = []
        name = ctypes.create_string_buffer(256)
        off, kind, aux = ctypes.c_long(), ctypes.c_int(), ctypes.c_int()
        dims = (ctypes.c_int * 4)()
        for i in range(L.stabnet_net_num_params(self._h)):
            _lib.call("stabnet_net_param_info", self._h, i, name, 256, ctypes.byref(off), ctypes.byref(kind), dims,
                      ctypes.byref(aux))
            self.table.append((name.value.decode(), off.value, kind.value, tuple(dims), aux.value))

    @property
    def handle(self):
        return self._h

    def __del__(self):
        try:
            if self._h:
                _lib.lib().stabnet_net_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- TF-layout dict <-> flat buffer ------------------------------------------------------------------
    def pack(self, params: dict) -> np.ndarray:
        flat = np.zeros(self.n_floats, np.float32)
        for name, off, kind, dims, aux in self.table:
            v = np.asarray(params[name], np.float32)
            if kind == KIND_CONV_W:                      # HWIO -> OHWI, Cin padded
                co, kh, kw, cp = dims
                assert v.shape == (kh, kw, aux, co), (name, v.shape, dims, aux)
                t = np.zeros((co, kh, kw, cp), np.float32)
                t[..., :aux] = np.transpose(v, (3, 0, 1, 2))
                v = t
            elif kind == KIND_FC_W:                      # [in,out] -> [out,in]
                assert v.shape == (dims[1], dims[0]), (name, v.shape, dims)
                v = np.ascontiguousarray(v.T)
            else:
                assert v.shape == (dims[0],), (name, v.shape, dims)
            flat[off:off + v.size] = v.reshape(-1)
        return flat

    def unpack(self, flat) -> dict:
        flat = np.asarray(flat, np.float32)
        out = {}
        for name, off, kind, dims, aux in self.table:
            if kind == KIND_CONV_W:
                co, kh, kw, cp = dims
                v = flat[off:off + co * kh * kw * cp].reshape(co, kh, kw, cp)[..., :aux]
                out[name] = np.ascontiguousarray(np.transpose(v, (1, 2, 3, 0)))
            elif kind == KIND_FC_W:
                out[name] = np.ascontiguousarray(flat[off:off + dims[0] * dims[1]].reshape(dims[0], dims[1]).T)
            else:
                out[name] = flat[off:off + dims[0]].copy()
        return out


class Regressor:
    """theta = Regressor(params)(x_tensor): resnet_v2_50 -> mean -> FC head in moving-average BN mode."""

    def __init__(self, params, N: int, H: int, W: int, cfg: Config = v2_93, device="cuda:0",
                 keep_activations: bool = False, bf16_operands=False, operand_mode=None):
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.StabnetError("Regressor needs a GPU device; there is no CPU fallback")
        # conv operand mode (include/stabnet_hip.h): 0 exact f32 MFMA (default), 1 bf16 operands (reduced precision), 2 / 3 split at
        # fragment-read time, 4 packed split (f32-level results on the bf16 matrix pipe).  `bf16_operands=True` is the older spelling of 1.
        mode = int(operand_mode) if operand_mode is not None else int(bf16_operands)
        self.operand_mode = mode
        packed_plan = (mode == 4 and not keep_activations)
        if packed_plan:                # split-K choices measured with the packed split kernels
            _lib.lib().stabnet_conv_tuning_profile(1)
        try:
            self.plan = NetPlan(N, H, W, cfg, keep_activations)
        finally:
            if packed_plan:
                _lib.lib().stabnet_conv_tuning_profile(0)
        if mode:
            _lib.call("stabnet_net_set_bf16_operands", self.plan.handle, mode)
        flat = params if isinstance(params, np.ndarray) and params.ndim == 1 else self.plan.pack(params)
        self.params = torch.from_numpy(np.ascontiguousarray(flat)).to(self.device)
        self.fold = torch.empty(int(_lib.lib().stabnet_net_fold_floats(self.plan.handle)), dtype=torch.float32, device=self.device)
        self.workspace = torch.empty(self.plan.workspace_bytes, dtype=torch.uint8, device=self.device)
        self._train = None                   # (plan, workspace, theta) of the is_training=True branch, made on first use
        self.refold()

    def refold(self):
        _lib.call("stabnet_net_fold_bn", self.plan.handle, ptr(self.params), ptr(self.fold), self.cfg.bn_eps,
                  stream_ptr(self.device), device=self.device)

    def forward(self, x_tensor: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        x = dev_f32(x_tensor, "x_tensor")
        p = self.plan
        assert tuple(x.shape) == (p.N, p.H, p.W, self.cfg.in_ch), "x_tensor %s != plan %s" % (
            tuple(x.shape), (p.N, p.H, p.W, self.cfg.in_ch))
        theta = out if out is not None else empty((p.N, self.cfg.n_theta), x)
        _lib.call("stabnet_backbone_fwd_infer", p.handle, ptr(self.params), ptr(self.fold), ptr(x), ptr(theta),
                  ptr(self.workspace), self.workspace.numel(), stream_ptr(self.device), 0, device=self.device)
        return theta

    __call__ = forward

    def forward_train(self, x_tensor: torch.Tensor) -> torch.Tensor:
        """get_resnet(..., is_training=True) (s_net_bundle_nobm.py:301): batch-statistics BN; the moving averages held in
        `self.params` are updated with cfg.bn_decay (slim UPDATE_OPS, :355-356).  Activations stay in the training
        workspace (what stabnet_tower_bwd consumes).  Call refold() before the next inference-mode forward."""
        x = dev_f32(x_tensor, "x_tensor")
        p = self.plan
        assert tuple(x.shape) == (p.N, p.H, p.W, self.cfg.in_ch), "x_tensor %s != plan %s" % (
            tuple(x.shape), (p.N, p.H, p.W, self.cfg.in_ch))
        if self._train is None:
            tp = NetPlan(p.N, p.H, p.W, self.cfg, keep_activations=True)
            nbytes = _lib.lib().stabnet_net_train_workspace_bytes(tp.handle)
            self._train = (tp, torch.empty(nbytes, dtype=torch.uint8, device=self.device))
        tp, ws = self._train
        theta = empty((p.N, self.cfg.n_theta), x)
        _lib.call("stabnet_tower_fwd_train", tp.handle, ptr(self.params), ptr(x), ptr(theta), ptr(ws), ws.numel(),
                  self.cfg.bn_eps, self.cfg.bn_decay, stream_ptr(self.device), 0, device=self.device)
        return theta

    def activation(self, name: str) -> torch.Tensor:
        """Debug tap (plan built with keep_activations=True): a view into the workspace, NHWC."""
        off = ctypes.c_long()
        dims = (ctypes.c_int * 4)()
        _lib.call("stabnet_net_activation_info", self.plan.handle, name.encode(), ctypes.byref(off), dims)
        n = dims[0] * dims[1] * dims[2] * dims[3]
        return self.workspace.view(torch.float32)[off.value:off.value + n].view(*dims)


def get_resnet(x_tensor, reuse=None, is_training=False, x_batch_size=None, *, regressor: Regressor):
    """s_net_bundle_nobm.py:250-264 -> (theta, id_loss, id2_loss); id2_loss = mean|theta| * id_mul (:263), and the
    reference returns it for both (`return theta, id2_loss, id2_loss`, :264).  is_training=True is the batch-statistics
    branch the reference builds at :301; False the moving-average branch of :302 (what deploy runs).  `reuse` and
    `x_batch_size` are graph-construction arguments of TF with no counterpart here (the batch is the plan's N)."""
    theta = regressor.forward_train(x_tensor) if is_training else regressor(x_tensor)
    cfg = regressor.cfg
    losses4 = empty((4,), theta)                 # {id2_loss, black_pos, distortion, consistency}; d_theta not wanted (NULL)
    _lib.call("stabnet_mesh_losses", ptr(theta), 0, theta.shape[0], cfg.grid_h, cfg.grid_w, cfg.do_crop_rate, cfg.id_mul,
              0.0, 0.0, 0.0, 0.0, 0.0, ptr(losses4), 0, stream_ptr(theta.device), device=theta.device)
    id2 = losses4[0]
    return theta, id2, id2
