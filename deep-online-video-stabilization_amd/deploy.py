"""Host side of the online loop (deploy_bundle.py:183-342, the network + feedback part): an on-device history ring
per stream and ONE C call per frame.  Video decode/encode and the colour remap stay with the caller."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from ._tensor import dev_f32, ptr, stream_ptr
from .config import Config, v2_93
from .regressor import Regressor


class Profiler:
    """Per-launch HIP-event records taken inside the library (bench.py roofline leg)."""

    def __init__(self, max_records: int = 65536, device=None):
        self._h = ctypes.c_void_p()
        self.overhead_ms = 0.0
        self.idle_pair_ms = 0.0
        self.offsets_us = {}
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        _lib.call("stabnet_prof_create", ctypes.byref(self._h), max_records)

    @property
    def handle(self):
        return self._h

    def reset(self):
        _lib.call("stabnet_prof_reset", self._h)

    def calibrate(self, n: int = 200) -> float:
        """Median duration (ms) of an event pair with NOTHING between on an idle stream = two hipEventRecords.  Around a kernel
        in a busy stream only about half of it is added to the kernel's own duration (the start event is processed while the
        previous kernel drains): measured against rocprofv3 on every conv kernel of a 720p frame, raw event time - rocprofv3
        time = 2.55 +- 0.1 us with an idle pair of 4.6 us.  The default correction is therefore HALF the idle pair; a
        per-kernel offset table (set_offsets: calibrated against the committed rocprofv3 averages of the same build by
        tools/profile_stamp.py) replaces it where it exists."""
        from ._tensor import stream_ptr
        self.reset()
        saved = (self.overhead_ms, self.offsets_us)
        self.overhead_ms, self.offsets_us = 0.0, {}
        for _ in range(n):
            _lib.call("stabnet_prof_record_empty", self._h, stream_ptr(self.device), device=self.device)
        ms = sorted(r[1] for r in self.records())
        self.reset()
        self.idle_pair_ms = ms[len(ms) // 2]
        self.overhead_ms, self.offsets_us = 0.5 * self.idle_pair_ms, saved[1]
        return self.overhead_ms

    def set_offsets(self, offsets_us: dict):
        """{kernel name: us to subtract from its raw event time} -- kernels not in the table get the default (calibrate())."""
        self.offsets_us = dict(offsets_us or {})

    def offset_ms(self, name: str) -> float:
        if name in self.offsets_us:
            return 1e-3 * self.offsets_us[name]
        return self.overhead_ms

    def records(self, raw: bool = False):
        """[(kernel name, ms, flops, bytes)] -- synchronises the device first.  ms = event time minus the kernel's offset
        (raw=True: the event time itself)."""
        torch.cuda.synchronize()
        L = _lib.lib()
        out = []
        kind, ms, fl, by = ctypes.c_int(), ctypes.c_float(), ctypes.c_double(), ctypes.c_double()
        for i in range(L.stabnet_prof_num_records(self._h)):
            _lib.call("stabnet_prof_record", self._h, i, ctypes.byref(kind), ctypes.byref(ms), ctypes.byref(fl),
                      ctypes.byref(by))
            name = L.stabnet_prof_kind_name(kind.value).decode()
            out.append((name, ms.value if raw else max(ms.value - self.offset_ms(name), 0.0), fl.value, by.value))
        return out

    def records_with_shapes(self):
        recs = self.records()
        shp = (ctypes.c_int * 4)()
        out = []
        for i, r in enumerate(recs):
            _lib.call("stabnet_prof_record_shape", self._h, i, shp)
            out.append(r + (tuple(shp),))
        return out

    def __del__(self):
        try:
            if self._h:
                _lib.lib().stabnet_prof_destroy(self._h)
                self._h = None
        except Exception:
            pass


class StabNetStream:
    """S independent video streams stabilised in lock-step on one GPU.

    step(cur) takes the next unstable frames [S,H,W] (train-normalised grey, [-0.5,0.5]) and returns the tensors the
    reference fetches at deploy_bundle.py:286 -- output_img, black_pix, Hs, x_map, y_map -- plus theta and the fed-back
    frame (img - black).  before_ch is accepted and ignored exactly like the reference (deploy_bundle.py:15,41): the
    ring depth is max(indices[1:])."""

    def __init__(self, params, H: int, W: int, cfg: Config = v2_93, streams: int = 1, device="cuda:0", refine: int = 1,
                 before_ch=None, use_graph: bool = False, bf16_operands: bool = False):
        self.cfg, self.H, self.W, self.S, self.refine = cfg, H, W, streams, refine
        self.reg = Regressor(params, streams, H, W, cfg, device, bf16_operands=bf16_operands)
        dev = self.reg.device
        self.lags = [i for i in cfg.indices[1:] if i > 0]
        self.depth = max(self.lags)
        self._lags_c = (ctypes.c_int * len(self.lags))(*self.lags)
        f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        self.frames_ring = f(streams, self.depth, H, W)
        self.masks_ring = f(streams, self.depth, H, W)
        self.theta = f(streams, cfg.n_theta)
        self.out_img = f(streams, H, W, 1)
        self.black = f(streams, H, W)
        self.x_map = f(streams, H, W, 1)
        self.y_map = f(streams, H, W, 1)
        self.Hs = f(streams, cfg.grid_h, cfg.grid_w, 9)
        self.frame_fb = f(streams, H, W)
        self.cur = f(streams, H, W)                       # fixed-address staging buffer of the current frame
        self.head_dev = torch.zeros(2, dtype=torch.int32, device=dev)     # {ring head, ticket}: on the device (graph replay)
        self.all_black = None                             # optional int32 [S,H,W]: += round(black) per refine pass (:291)
        self.use_graph = use_graph
        self._graph = None
        self.started = False

    @property
    def head(self) -> int:
        """Ring slot the NEXT frame's push writes (host read-back; synchronises)."""
        return int(self.head_dev[0].item())

    def start(self, first_frame: torch.Tensor):
        first = dev_f32(first_frame, "first_frame").reshape(self.S, self.H, self.W)
        _lib.call("stabnet_ring_init", ptr(self.frames_ring), ptr(self.masks_ring), ptr(first), self.S, self.depth,
                  self.H, self.W, stream_ptr(self.reg.device), device=self.reg.device)
        self.head_dev.zero_()
        if self.all_black is not None:
            self.all_black.zero_()
        self.started = True

    def track_black(self, enable: bool = True):
        """Accumulate all_black (deploy_bundle.py:234,291) on the device for the crop search; reset by start()."""
        self.all_black = (torch.zeros((self.S, self.H, self.W), dtype=torch.int32, device=self.reg.device) if enable else None)
        self._graph = None
        return self.all_black

    def _enqueue(self, prof=None):
        r = self.reg
        _lib.call("stabnet_deploy_frame", r.plan.handle, ptr(r.params), ptr(r.fold), ptr(self.frames_ring),
                  ptr(self.masks_ring), self.depth, ptr(self.head_dev), self._lags_c, len(self.lags), ptr(self.cur),
                  self.refine, self.cfg.grid_h, self.cfg.grid_w, self.cfg.do_crop_rate, ptr(self.theta), ptr(self.out_img),
                  ptr(self.black), ptr(self.x_map), ptr(self.y_map), ptr(self.Hs), ptr(self.frame_fb), ptr(self.all_black),
                  ptr(r.workspace), r.workspace.numel(), stream_ptr(r.device), prof.handle if prof is not None else 0,
                  device=r.device)

    def step(self, cur: torch.Tensor, prof: Profiler = None):
        if not self.started:
            raise _lib.StabnetError("StabNetStream.step before start(first_frame)")
        self.cur.copy_(dev_f32(cur, "cur").reshape(self.S, self.H, self.W))            # D2D into the fixed buffer
        if self.use_graph and prof is None:
            if self._graph is None:
                # one frame = ~95 launches with fixed arguments: capture once, replay per frame (hipGraph)
                self._enqueue()                  # this frame runs eagerly (also loads modules / sets kernel attributes)
                torch.cuda.synchronize()
                try:
                    with torch.cuda.device(self.reg.device):
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g):    # capture records only; nothing executes here
                            self._enqueue()
                    self._graph = g
                except Exception as e:               # capture unsupported on this runtime: stay eager, say so once
                    import sys
                    print("StabNetStream: hipGraph capture failed (%s); continuing without a graph" % e, file=sys.stderr)
                    self.use_graph = False
            else:
                self._graph.replay()
        else:
            self._enqueue(prof)
        return {"output": self.out_img, "black_pix": self.black, "Hs": self.Hs, "x_map": self.x_map,
                "y_map": self.y_map, "theta": self.theta, "frame": self.frame_fb}
