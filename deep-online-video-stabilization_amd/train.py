"""Host side of the training step (train_bundle_nobm.py:107-160,216-236,327-346): two siamese towers sharing the
weights, the temporal loss through the flow sampler, the loss schedule gates, Adam with the staircase learning rate,
and -- new, the reference is single-device -- data parallelism: one process per GPU, samples sharded across
ranks, local BN statistics, ONE 121.6 MB gradient per step (both towers accumulate into the same buffer) all-reduced over
RCCL (torch.distributed "nccl") in reverse layer order: the two towers run layer by layer in LOCKSTEP (one launch per
batch-statistics reduction for both, dgrad weights packed once), their backward in four stages (FC head + block4, block3,
block2, block1 + stem), and each stage's parameter bucket goes to the communication stream as soon as that stage is
enqueued, so the collective of the big late layers overlaps the backward of the early ones.
Every tensor op is a C-ABI kernel; torch allocates, holds pointers, owns the streams and the collective.  The step is
reproducible bit for bit (no order-dependent float atomics anywhere on the path)."""
from __future__ import annotations

import os

import ctypes

import numpy as np
import torch

from . import _lib, parallel, train_ops, warp
from ._tensor import dev_f32, ptr, stream_ptr
from .config import Config, v2_93
from .regressor import KIND_CONV_W, NetPlan

RET_KEYS = ("error", "black_pos", "black_pix", "theta_loss", "grid_theta_loss", "black_loss", "distortion_loss",
            "consistency_loss", "feature_loss", "mask", "matches", "img_loss", "regu_loss", "x_tensor", "use_theta_only",
            "y", "output", "total_loss", "use_theta_loss", "use_black_loss", "stable_warpped")


def loss_gates(i: int, cfg: Config):
    """Per-step scalar gates fed as placeholders (train_bundle_nobm.py:219-236)."""
    use_theta = 0 if i > cfg.no_theta_iter else 1
    if i <= cfg.do_theta_10_iter:
        use_theta = 10
    return {"use_theta_loss": use_theta, "use_temp_loss": 1 if i >= cfg.do_temp_loss_iter else 0,
            "use_black_loss": 1 if i >= cfg.do_black_loss_iter else 0,
            "use_theta_only": 1 if i <= cfg.do_theta_only_iter else 0}


def learning_rate(step: int, cfg: Config) -> float:
    """tf.train.exponential_decay(..., decay_rate=0.1, staircase=True) (train_bundle_nobm.py:155-158), in float32 as TF
    computes it: lr0 * pow(0.1, floor(step / step_size))."""
    p = np.float32(np.floor(np.float32(step) / np.float32(cfg.step_size)))
    return float(np.float32(cfg.initial_learning_rate) * np.float32(np.power(np.float32(0.1), p)))


class Trainer:
    def __init__(self, params, N: int, H: int, W: int, cfg: Config = v2_93, device="cuda:0", process_group=None,
                 world_size: int = 1, force_comm: bool = False, split_operands=None):
        """force_comm: run the communication path (buckets on the communication stream, wait_stream joins) even for a
        one-rank group -- the sum over one rank is the identity, so the step must equal the no-group step bit for bit; that
        is how the RCCL path is exercised on a one-GPU box (tests/test_rccl_gpu.py)."""
        self.cfg, self.N, self.H, self.W = cfg, N, H, W
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.StabnetError("Trainer needs a GPU device; there is no CPU fallback")
        self.plan = NetPlan(N, H, W, cfg, keep_activations=True)
        # split_operands: the launches of the step whose B operand is a weight tensor -- the prologue-carrying 1x1 forward pairs and
        # the stride-1 dgrad launches -- run the packed split kernels (float32 operands as exact sums of three bf16 terms on the bf16
        # matrix pipe, f32 accumulate: include/stabnet_hip.h, operand mode 4) on images of the forward weights and of the re-packed
        # dgrad weights written once per step (two launches).  None = the STABNET_TRAIN_SPLIT environment switch, default OFF.
        # Measured at 8 pairs / 288x512: the converted launches are 10-24 % faster (1x1 forward pairs 57.7 ->
        # 51.9 us, dgrad 49.7 -> 38.7 us and 104.0 -> 77.9 us: -1.0 ms of 15.2 per step) and a 10-step run shows 527 -> 540
        # pairs/s -- but over 200 steps it is 526.1 against 527.2: the chip is power-limited, and once the clocks settle every
        # f32-MFMA launch beside the packed ones (wgrad, the 3x3 forward pairs: 60 % of the matrix time, neither can read a weight
        # image) runs 4-8 % slower.  It pays at larger batches (16 / 32 / 64 pairs per GPU: +2 / +3 / +6 %).
        if split_operands is None:
            import os
            split_operands = os.environ.get("STABNET_TRAIN_SPLIT", "0") == "1"
        self.split_operands = bool(split_operands)
        if self.split_operands:
            _lib.call("stabnet_net_set_bf16_operands", self.plan.handle, 4)
        flat = params if isinstance(params, np.ndarray) and params.ndim == 1 else self.plan.pack(params)
        dev = self.device
        self.params = torch.from_numpy(np.ascontiguousarray(flat)).to(dev)
        nt = self.plan.n_trainable
        self.nt = nt
        self.grads = torch.zeros(nt, dtype=torch.float32, device=dev)        # ONE buffer: both towers accumulate into it
        self.adam_m = torch.zeros(nt, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(nt, dtype=torch.float32, device=dev)
        L = _lib.lib()
        self.ws_bytes = L.stabnet_net_train_workspace_bytes(self.plan.handle)
        self.ws = [torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.theta = [torch.empty((N, cfg.n_theta), dtype=torch.float32, device=dev) for _ in range(2)]
        # L2-regulariser segments: slim conv weights 1e-4, fc_weights / fc_bias 2e-4 (s_net_bundle_nobm.py:324-325)
        offs, lens, coefs = [], [], []
        for name, off, kind, dims, aux in self.plan.table:
            if kind == KIND_CONV_W:
                offs.append(off); lens.append(int(np.prod(dims))); coefs.append(cfg.weight_decay_conv)
            elif name in ("fc/fc_weights", "fc/fc_bias"):
                n = dims[0] * dims[1] if name == "fc/fc_weights" else dims[0]
                offs.append(off); lens.append(n); coefs.append(cfg.weight_decay_fc)
        self.seg_off = torch.tensor(offs, dtype=torch.int64, device=dev)
        self.seg_len = torch.tensor(lens, dtype=torch.int64, device=dev)
        self.seg_coef = torch.tensor(coefs, dtype=torch.float32, device=dev)
        self.regu_val = torch.zeros(1, dtype=torch.float32, device=dev)
        self.wd_ws = torch.empty(64 * len(offs), dtype=torch.float32, device=dev)
        self.global_step = 0
        self.pg = process_group
        self.world = world_size
        self.comm = world_size > 1 or (force_comm and process_group is not None)
        if self.comm and os.environ.get("STABNET_COMM_RESERVED_CUS"):
            # CUs the persistent conv grids leave free for the collective's kernels.  Default: none -- measured with a stand-in
            # kernel (tools/comm_proxy.py) a 32-workgroup kernel on a second stream gets its CU slots within
            # 1.5 us beside the backward without any reservation, and reserving 16 / 32 CUs costs the step 1.1 / 2.6 %.
            _lib.lib().stabnet_conv_reserve_cus(int(os.environ["STABNET_COMM_RESERVED_CUS"]))
        self.comm_stream = torch.cuda.Stream(device=dev) if self.comm else None
        self.last = None
        self.prof = None                                        # deploy.Profiler: per-launch HIP events (bench only)
        # gradient buckets in the order backward completes them (reverse layer order) + the BN gamma/beta sections
        self.n_stages = L.stabnet_net_num_grad_stages()
        lo, hi = ctypes.c_long(), ctypes.c_long()
        self.buckets = []
        for k in range(self.n_stages):
            _lib.call("stabnet_net_grad_bucket", self.plan.handle, k, ctypes.byref(lo), ctypes.byref(hi))
            self.buckets.append((lo.value, hi.value))
        _lib.call("stabnet_net_bn_grad_range", self.plan.handle, ctypes.byref(lo), ctypes.byref(hi))
        self.bn_bucket = (lo.value, hi.value)
        assert sorted(self.buckets + [self.bn_bucket])[0][0] == 0 and sum(b - a for a, b in self.buckets + [self.bn_bucket]) == nt
        self.comm_timing = None               # bench: list of (start, end, bytes) on the comm stream, one per bucket
        self.compute_done = []                # bench: one event per step on the compute stream, just before the join

    # ------------------------------------------------------------------------------------------------------
    def _towers_fwd(self, x1, x2):
        """Both siamese towers layer by layer in lockstep (one batch-statistics reduction launch per layer for both)."""
        _lib.call("stabnet_towers_fwd_train", self.plan.handle, ptr(self.params), ptr(x1), ptr(x2), ptr(self.theta[0]),
                  ptr(self.theta[1]), ptr(self.ws[0]), ptr(self.ws[1]), self.ws_bytes, self.cfg.bn_eps, self.cfg.bn_decay,
                  stream_ptr(self.device), self.prof.handle if self.prof is not None else 0, device=self.device)
        return self.theta

    def _towers_bwd(self, d_theta1, d_theta2):
        """Backward of both towers in lockstep into the ONE gradient buffer, stage by stage (FC head + block4, block3, block2,
        block1 + stem); with world > 1 each finished bucket is handed to the communication stream (reverse layer order), so
        the collective of the big late layers overlaps the backward of the early ones."""
        prof = self.prof.handle if self.prof is not None else 0
        for stage in range(self.n_stages):
            _lib.call("stabnet_towers_bwd_stage", self.plan.handle, ptr(self.params), ptr(d_theta1), ptr(d_theta2),
                      ptr(self.grads), ptr(self.ws[0]), ptr(self.ws[1]), self.ws_bytes, stage, stream_ptr(self.device), prof,
                      device=self.device)
            if self.comm:
                self._allreduce_async(*self.buckets[stage])
        if self.comm:
            self._allreduce_async(*self.bn_bucket)

    def _allreduce_async(self, lo: int, hi: int):
        """Sum grads[lo:hi] over ranks on the communication stream, ordered after everything enqueued so far."""
        import torch.distributed as dist
        cs = self.comm_stream
        cs.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(cs):
            if self.comm_timing is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cs)
            # xGMI is point-to-point, rings are per-link bound -> one large message per bucket, not many small ones
            dist.all_reduce(self.grads[lo:hi], group=self.pg)
            if self.comm_timing is not None:
                e1.record(cs)
                self.comm_timing.append((e0, e1, (hi - lo) * 4))

    def grad_flat(self) -> torch.Tensor:
        """The step's gradient (both towers; summed over ranks when world > 1), trainable layout."""
        return self.grads

    def forward_backward(self, batch: dict, gates: dict = None, apply_update: bool = True):
        """One optimiser step on a batch dict (x1,y1,x2,y2,flow,matches1,mask1,matches2,mask2), device tensors.
        Returns a dict of device scalars/tensors (read them on the host only when logging)."""
        cfg, N, H, W = self.cfg, self.N, self.H, self.W
        g = gates if gates is not None else loss_gates(self.global_step, cfg)
        to = float(g["use_theta_only"])
        live = 1.0 - to
        cur = 2 * cfg.before_ch if cfg.input_mask else cfg.before_ch
        self.grads.zero_()
        self.regu_val.zero_()
        towers = []
        xs = [dev_f32(batch["x1"]), dev_f32(batch["x2"])]
        thetas = self._towers_fwd(xs[0], xs[1])
        for k in (0, 1):
            x, theta = xs[k], thetas[k]
            frame = warp.slice_channel(x, cur)                                    # x = x_tensor[..., 12:13], s_net:281
            r = warp.warp_from_theta(frame, theta, cfg)
            r["frame"] = frame
            towers.append(r)
        flow = dev_f32(batch["flow"])
        fx, fy = warp.slice_channel(flow, 0), warp.slice_channel(flow, 1)
        # ---- loss values (forward)
        img_sums = [train_ops.masked_mse_sums(t["output"], dev_f32(batch["y" + s]), t["black_pix"])
                    for t, s in zip(towers, ("1", "2"))]
        o2w = warp.interpolate(towers[1]["output"], fx, fy)
        nb2w = warp.interpolate(train_ops.axpb(towers[1]["black_pix"], -1.0, 1.0).reshape(N, H, W, 1), fx, fy)
        t_sums = train_ops.masked_mse_sums(towers[0]["output"], o2w, towers[0]["black_pix"], nb2w)
        feats = [train_ops.feature_loss(dev_f32(batch["matches" + s]), dev_f32(batch["mask" + s]), t["x_map"], t["y_map"],
                                        live * cfg.feature_mul / N, want_grad=True, want_warped=True)
                 for t, s in zip(towers, ("1", "2"))]
        # ---- d(total)/d(out_k)
        c_img = live * cfg.img_mul / cfg.batch_size
        c_tmp = cfg.temp_mul * float(g["use_temp_loss"]) / cfg.batch_size
        g_out = []
        for k, s in enumerate(("1", "2")):
            ga, _ = train_ops.masked_mse_grad(towers[k]["output"], dev_f32(batch["y" + s]), towers[k]["black_pix"], None,
                                              img_sums[k], c_img)
            g_out.append(ga)
        _, g_o2w = train_ops.masked_mse_grad(towers[0]["output"], o2w, towers[0]["black_pix"], nb2w, t_sums, c_tmp,
                                             ga=g_out[0], accumulate_a=True, want_gb=True)
        train_ops.interp_bwd(fx, fy, g_o2w, d_im=g_out[1])
        # ---- through the warp and the mesh losses to d theta of both towers, then the joint staged backward
        w_id = cfg.theta_mul + cfg.grid_theta_mul
        mesh = [None, None]
        d_thetas = [None, None]
        for k in (0, 1):
            t = towers[k]
            d_pts2 = train_ops.transformer_bwd(t["pts2"], t["Hs"], t["frame"], t["x_map"], t["y_map"], g_out[k],
                                               feats[k][1], feats[k][2], cfg, dmap_scale=feats[k][4])
            losses4, d_theta = train_ops.mesh_losses(self.theta[k], d_pts2, cfg, w_id, live * cfg.distortion_mul,
                                                     live * cfg.consistency_mul, float(g["use_black_loss"]),
                                                     live * cfg.black_mul)
            mesh[k] = losses4
            d_thetas[k] = d_theta
        self._towers_bwd(d_thetas[0], d_thetas[1])
        if self.comm:
            if self.comm_timing is not None:          # bench: when did backward itself finish (vs the last bucket's end)?
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(self.device))
                self.compute_done.append(ev)
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        # regu_loss is counted once per tower (train_bundle_nobm.py:142): gradient coefficient 2 * regu_mul * live
        # (times world: the summed gradient is divided by world in the Adam kernel, the regulariser is not a rank sum)
        _lib.call("stabnet_weight_decay", ptr(self.params), ptr(self.grads), ptr(self.seg_off), ptr(self.seg_len),
                  ptr(self.seg_coef), self.seg_off.numel(), 2.0 * cfg.regu_mul * live * self.world, ptr(self.regu_val),
                  ptr(self.wd_ws), stream_ptr(self.device), device=self.device)
        self.global_step += 1
        if apply_update:
            lr = learning_rate(self.global_step - 1, cfg)
            _lib.call("stabnet_adam_step", ptr(self.params), ptr(self.grads), 0, ptr(self.adam_m),
                      ptr(self.adam_v), self.nt, lr, 0.9, 0.999, 1e-8, self.global_step, 1.0 / self.world,
                      stream_ptr(self.device), device=self.device)
        self.last = {"towers": towers, "mesh": mesh, "img_sums": img_sums, "t_sums": t_sums, "feats": feats, "gates": g,
                     "batch": batch}
        return self.last

    # ------------------------------------------------------------------------------------------------------
    def losses(self) -> dict:
        """Host-side readout of the last step's loss terms with the reference's `ret` scaling (s_net_bundle_nobm.py:361-385,
        train_bundle_nobm.py:142-153).  Synchronises."""
        cfg, L = self.cfg, self.last
        g = L["gates"]
        live = 1.0 - float(g["use_theta_only"])
        regu = float(self.regu_val.item())
        out = {"regu_loss": 2 * regu * cfg.regu_mul}
        tot = 0.0
        for k in (0, 1):
            m = L["mesh"][k].cpu().numpy().astype(np.float64)
            s = L["img_sums"][k].cpu().numpy().astype(np.float64)
            img = float((s[:, 0] / (s[:, 1] + 1e-8)).sum() / cfg.batch_size)
            feat = float(L["feats"][k][0].mean().item())
            t = {"theta_loss": m[0] * cfg.theta_mul, "grid_theta_loss": m[0] * cfg.grid_theta_mul,
                 "black_loss": m[1] * cfg.black_mul, "distortion_loss": m[2] * cfg.distortion_mul,
                 "consistency_loss": m[3] * cfg.consistency_mul, "feature_loss": feat * cfg.feature_mul,
                 "img_loss": img * cfg.img_mul}
            t["total_loss"] = t["theta_loss"] + t["grid_theta_loss"] + live * (
                t["img_loss"] + regu * cfg.regu_mul + t["black_loss"] + t["distortion_loss"] + t["consistency_loss"]
                + t["feature_loss"])
            tot += t["total_loss"]
            for kk, v in t.items():
                out[kk] = out.get(kk, 0.0) + v if kk != "total_loss" else out.get(kk, 0.0)
            out["tower%d" % (k + 1)] = t
        s = L["t_sums"].cpu().numpy().astype(np.float64)
        temp = float((s[:, 0] / (s[:, 1] + 1e-8)).sum() / cfg.batch_size * float(g["use_temp_loss"]))
        out["temp_loss"] = temp * cfg.temp_mul
        out["total_loss"] = tot + temp * cfg.temp_mul
        return out

    def ret(self, k: int) -> dict:
        """The reference's per-tower `ret` dict (s_net_bundle_nobm.py:361-385) for tower k (0 or 1) of the last step."""
        L, cfg = self.last, self.cfg
        t = L["towers"][k]
        s = ("1", "2")[k]
        lo = self.losses()["tower%d" % (k + 1)]
        y = dev_f32(L["batch"]["y" + s])
        d = {key: None for key in RET_KEYS}
        d.update({"error": (t["output"] - y).abs(), "black_pix": t["black_pix"].reshape(self.N, self.H, self.W, 1),
                  "output": t["output"], "x_tensor": L["batch"]["x" + s], "y": y, "mask": L["batch"]["mask" + s],
                  "matches": L["batch"]["matches" + s], "stable_warpped": L["feats"][k][3],
                  "use_theta_only": L["gates"]["use_theta_only"], "use_theta_loss": L["gates"]["use_theta_loss"],
                  "use_black_loss": L["gates"]["use_black_loss"],
                  "black_pos": torch.zeros((self.N, cfg.grid_h * cfg.grid_w * 8), device=self.device)})
        d.update({k2: lo[k2] for k2 in ("theta_loss", "grid_theta_loss", "black_loss", "distortion_loss",
                                        "consistency_loss", "feature_loss", "img_loss", "total_loss")})
        d["regu_loss"] = float(self.regu_val.item()) * cfg.regu_mul
        return d

    def state_dict(self) -> dict:
        """Everything a resume needs.  Data parallel: trainables, Adam moments and the step are identical on every rank
        (same summed gradient, same update); the BN MOVING statistics are NOT -- every rank tracks its own shard's batch
        statistics (local BN) -- so call sync_moving_stats() first to checkpoint their rank mean."""
        return {"params": self.params.cpu().numpy(), "adam_m": self.adam_m.cpu().numpy(), "adam_v": self.adam_v.cpu().numpy(),
                "global_step": self.global_step}

    def sync_moving_stats(self):
        """Average the BN moving means / variances over ranks in place (no-op for one process).  The moving statistics
        are exponential averages of per-shard batch moments; their rank mean is the average over the global batch's
        shards, which is what a single-process run with `world` x more steps of the same decay would track in expectation.
        Done at checkpoint time only (train_bundle_nobm.py:271-272), never inside the step."""
        if self.world <= 1:
            return
        import torch.distributed as dist
        tail = self.params[self.nt:]
        dist.all_reduce(tail, group=self.pg)
        tail.div_(self.world)

    def load_state_dict(self, sd: dict):
        self.params.copy_(torch.from_numpy(sd["params"]))
        self.adam_m.copy_(torch.from_numpy(sd["adam_m"]))
        self.adam_v.copy_(torch.from_numpy(sd["adam_v"]))
        self.global_step = int(sd["global_step"])
