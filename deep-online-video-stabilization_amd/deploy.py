"""Host side of the online loop (deploy_bundle.py:183-342, the network + feedback part): an on-device history ring
per stream and ONE C call per frame (StabNetStream); ClipPipeline drives it from a clip in HOST memory with the PCIe copies
of neighbouring frames overlapped and the colour remap (warpRevBundle2) on the device.  Video decode/encode stay with the caller."""
from __future__ import annotations

import ctypes
import time

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
                 before_ch=None, use_graph: bool = False, bf16_operands=False, operand_mode=None):
        """operand_mode: conv operand mode of the regressor (regressor.Regressor; 4 = packed split kernels, what bench.py and
        deploy_bundle.py run; default 0 = exact f32 MFMA); bf16_operands=True is the older spelling of mode 1."""
        self.cfg, self.H, self.W, self.S, self.refine = cfg, H, W, streams, refine
        self.reg = Regressor(params, streams, H, W, cfg, device, bf16_operands=bf16_operands, operand_mode=operand_mode)
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

    def _enqueue(self, prof=None, cur=None):
        """One frame on the current stream.  cur: the frame to read instead of the fixed staging buffer self.cur (ClipPipeline hands
        its upload slot over directly)."""
        r = self.reg
        _lib.call("stabnet_deploy_frame", r.plan.handle, ptr(r.params), ptr(r.fold), ptr(self.frames_ring),
                  ptr(self.masks_ring), self.depth, ptr(self.head_dev), self._lags_c, len(self.lags), ptr(self.cur if cur is None else cur),
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


class ClipPipeline:
    """A clip that lives in HOST memory through one StabNetStream, with upload, frame and download on three HIP streams.

    The reference's loop (deploy_bundle.py:244-342) is strictly serial per frame: feed the frame, `sess.run`, `cv2.remap` the colour
    frame, write it.  The frame itself is recurrent (frame t reads what frame t-1 fed back through the ring), so frames cannot overlap
    each other -- but their PCIe traffic can hide behind the neighbours' compute: while frame t is on the compute stream, frame t+1
    (grey float32 + colour uint8, 6.4 MB at 720p) is uploaded from pinned memory on a second stream and the results of frame t-1
    (stabilised colour frame + the network's grey output, 3.7 MB) are downloaded on a third.  `slots` (>= 2) staging buffers per
    direction; events order the three streams, the host only blocks when it reuses a slot.  Results are the same bytes the serial
    loop produces (tests/test_pipeline_gpu.py).

    run(grey, bgr=None, sink=None, maps=False): grey = sequence of host float32 [H,W] frames in the training normalisation, frame 0 seeds
    the ring (deploy_bundle.py:206-222); bgr = matching uint8 [H,W,3] frames or None.  Per processed frame t >= 1 the host gets
    {"t", "output" uint8 [H,W], "bgr" uint8 [H,W,3] (if bgr), "x_map"/"y_map"/"black" (if maps)}: passed to `sink` as views of pinned
    staging memory that stay valid until the sink returns, or -- without a sink -- copied and returned as a list."""

    def __init__(self, stream: StabNetStream, colour: bool = True, slots: int = 3, rate: int = 4):
        if stream.S != 1:
            raise _lib.StabnetError("ClipPipeline: one video stream per pipeline")
        if slots < 2:
            raise _lib.StabnetError("ClipPipeline: slots must be >= 2")
        self.st, self.colour, self.slots, self.rate = stream, colour, slots, rate
        dev = stream.reg.device
        H, W = stream.H, stream.W
        self.dev = dev
        pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)
        on = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        self.h_grey = [pin((H, W), torch.float32) for _ in range(slots)]
        self.d_grey = [on((1, H, W), torch.float32) for _ in range(slots)]
        self.h_out = [pin((H, W), torch.uint8) for _ in range(slots)]
        self.d_out = [on((H, W), torch.uint8) for _ in range(slots)]
        if colour:
            self.h_bgr = [pin((H, W, 3), torch.uint8) for _ in range(slots)]
            self.d_bgr = [on((1, H, W, 3), torch.uint8) for _ in range(slots)]
            self.h_warp = [pin((H, W, 3), torch.uint8) for _ in range(slots)]
            self.d_warp = [on((1, H, W, 3), torch.uint8) for _ in range(slots)]
            self.remap_ws = on((2 * (H // rate) * (W // rate),), torch.float32)
        self.h_maps = None
        self.use_graph = True
        self._graphs = {}
        self.s_in, self.s_run, self.s_out = (torch.cuda.Stream(device=dev) for _ in range(3))
        self.ev_up = [torch.cuda.Event() for _ in range(slots)]
        self.ev_run = [torch.cuda.Event() for _ in range(slots)]
        self.ev_down = [torch.cuda.Event() for _ in range(slots)]

    def _frame(self, k: int, maps: bool):
        """Everything frame-shaped of slot k on the current stream: the frame, its results into the slot's buffers."""
        st, H, W = self.st, self.st.H, self.st.W
        st._enqueue(cur=self.d_grey[k])                      # the frame reads the upload slot itself: no staging copy
        # cvt_train2img (deploy_bundle.py:75)
        _lib.call("stabnet_cvt_train2img", ptr(st.out_img), ptr(self.d_out[k]), H * W, stream_ptr(self.dev), device=self.dev)
        if self.colour:
            _lib.call("stabnet_warp_rev_bundle2", ptr(self.d_bgr[k]), ptr(st.x_map), ptr(st.y_map), 1, H, W, 3, self.rate,
                      ptr(self.d_warp[k]), ptr(self.remap_ws), 0, 0, stream_ptr(self.dev), device=self.dev)
        if maps:
            self.d_maps[0][k].copy_(st.x_map.view(H, W)); self.d_maps[1][k].copy_(st.y_map.view(H, W))
            self.d_maps[2][k].copy_(st.black.view(H, W))

    def _capture(self, k: int, maps: bool):
        try:
            with torch.cuda.device(self.dev):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._frame(k, maps)
            return g
        except Exception as e:                       # capture unsupported on this runtime: stay eager, say so once
            import sys
            print("ClipPipeline: hipGraph capture failed (%s); continuing without graphs" % e, file=sys.stderr)
            self.use_graph = False
            return None

    def _result(self, slot: int, t: int, maps: bool):
        r = {"t": t, "output": self.h_out[slot].numpy()}
        if self.colour:
            r["bgr"] = self.h_warp[slot].numpy()
        if maps:
            r["x_map"], r["y_map"], r["black"] = (m[slot].numpy() for m in self.h_maps)
        return r

    def run(self, grey, bgr=None, sink=None, maps: bool = False):
        st, H, W, K = self.st, self.st.H, self.st.W, self.slots
        if self.colour and bgr is None:
            raise _lib.StabnetError("ClipPipeline.run: this pipeline was built with colour=True, bgr frames are required")
        if maps and self.h_maps is None:
            self.h_maps = tuple([torch.empty((H, W), dtype=dt, pin_memory=True) for _ in range(K)]
                                for dt in (torch.float32, torch.float32, torch.uint8))
            self.d_maps = tuple([torch.empty((H, W), dtype=dt, device=self.dev) for _ in range(K)]
                                for dt in (torch.float32, torch.float32, torch.uint8))
        results = []
        emit = sink if sink is not None else (lambda r: results.append({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in r.items()}))
        n = len(grey)
        self.host_wait_s = 0.0                                # time the host spent blocked on a slot: ~0 means the HOST is the bound
        torch.cuda.synchronize(self.dev)
        with torch.cuda.stream(self.s_run):
            first = torch.from_numpy(np.ascontiguousarray(grey[0], dtype=np.float32)).to(self.dev)
            st.start(first[None])
        pending = [None] * K                                  # frame number whose results sit in (or are on their way to) slot k
        for t in range(1, n):
            k = t % K
            if pending[k] is not None:                        # slot reuse: its download must have landed; hand the frame over
                w0 = time.perf_counter()
                self.ev_down[k].synchronize()
                self.host_wait_s += time.perf_counter() - w0
                emit(self._result(k, pending[k], maps))
                pending[k] = None
            self.h_grey[k].numpy()[...] = grey[t]
            if self.colour:
                self.h_bgr[k].numpy()[...] = bgr[t]
            with torch.cuda.stream(self.s_in):
                self.d_grey[k].copy_(self.h_grey[k].view(1, H, W), non_blocking=True)
                if self.colour:
                    self.d_bgr[k].copy_(self.h_bgr[k].view(1, H, W, 3), non_blocking=True)
                self.ev_up[k].record(self.s_in)
            with torch.cuda.stream(self.s_run):
                self.s_run.wait_event(self.ev_up[k])
                key = (k, maps, ptr(st.all_black))
                g = self._graphs.get(key)
                if g is not None:
                    g.replay()
                else:
                    # first use of this slot: the frame runs eagerly, then the same launches are RECORDED (nothing executes during a
                    # capture) into the slot's own hipGraph: in steady state a frame is one graph launch per slot, no eager kernels
                    self._frame(k, maps)
                    if self.use_graph:
                        self._graphs[key] = self._capture(k, maps)
                self.ev_run[k].record(self.s_run)
            with torch.cuda.stream(self.s_out):
                self.s_out.wait_event(self.ev_run[k])
                self.h_out[k].copy_(self.d_out[k], non_blocking=True)
                if self.colour:
                    self.h_warp[k].copy_(self.d_warp[k].view(H, W, 3), non_blocking=True)
                if maps:
                    for h, d in zip(self.h_maps, self.d_maps):
                        h[k].copy_(d[k], non_blocking=True)
                self.ev_down[k].record(self.s_out)
            pending[k] = t                                    # (the slot's next upload follows the host's wait on ev_down[k])
        order = sorted((p, k) for k, p in enumerate(pending) if p is not None)
        for p, k in order:
            self.ev_down[k].synchronize()
            emit(self._result(k, p, maps))
        torch.cuda.synchronize(self.dev)
        return results if sink is None else None
