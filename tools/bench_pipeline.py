#!/usr/bin/env python3
"""Host-to-host rate of the deploy loop ("Host buffers"): a 720p clip that lives in HOST memory (grey float32 +
colour uint8 frames) in, stabilised colour frame + the network's grey output back in host memory, per frame.
  serial    the loop as the reference writes it (deploy_bundle.py:244-342) and as ./deploy_bundle.py runs it: upload, frame, remap,
            download, one after the other with the host waiting in between
  pipeline  stabnet_amd.deploy.ClipPipeline: the same work with upload / frame / download on three HIP streams (pinned staging)
  resident  the frame alone, inputs already in HBM (= what bench.py reports as `value`)
One JSON object on stdout.   python tools/bench_pipeline.py [--frames 300] [--height 720 --width 1280]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stabnet_amd import synthetic, warp
from stabnet_amd.config import Config
from stabnet_amd.deploy import ClipPipeline, StabNetStream

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--height", type=int, default=720)
ap.add_argument("--width", type=int, default=1280)
ap.add_argument("--slots", type=int, default=3)
a = ap.parse_args()
H, W, T = a.height, a.width, a.frames
dev = torch.device("cuda", 0)
cfg = Config(height=H, width=W)
params = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
base = synthetic.make_clip(H, W, 40, seed=1234).astype(np.float32)
grey = np.ascontiguousarray(base[np.arange(T) % len(base)])                       # T host frames
bgr = np.ascontiguousarray(np.repeat(((grey + 0.5) * 255).clip(0, 255).astype(np.uint8)[..., None], 3, axis=3))
out = {"device": torch.cuda.get_device_name(0), "height": H, "width": W, "frames": T - 1,
       "host_bytes_per_frame": {"up": H * W * 4 + H * W * 3, "down": H * W * 3 + H * W}}


def serial():
    st = StabNetStream(params, H, W, cfg, device=dev, use_graph=True)
    st.start(torch.from_numpy(grey[0][None]).to(dev))
    for t in range(1, 4):                                                        # graph capture + warm-up
        st.step(torch.from_numpy(grey[t][None]).to(dev))
    torch.cuda.synchronize()
    st.start(torch.from_numpy(grey[0][None]).to(dev))
    t0 = time.perf_counter()
    for t in range(1, T):
        cur = torch.from_numpy(grey[t][None]).to(dev)
        r = st.step(cur)
        c = warp.warpRevBundle2(torch.from_numpy(bgr[t]).to(dev), r["x_map"], r["y_map"]).cpu().numpy()
        o = ((r["output"][0, :, :, 0].cpu().numpy() + 0.5) * 255).clip(0, 255).astype(np.uint8)
    dt = time.perf_counter() - t0
    return (T - 1) / dt, (c, o)


def pipeline(consume=True):
    st = StabNetStream(params, H, W, cfg, device=dev, use_graph=True)
    pipe = ClipPipeline(st, colour=True, slots=a.slots)
    pipe.run(grey[:120], bgr[:120], sink=lambda r: None)                         # graph capture + warm-up (clocks: a run that follows the
                                                                                 # half-idle serial loop is ~6 % slower for its first 0.3 s)
    got_c, got_o = np.zeros((T, H, W, 3), np.uint8), np.zeros((T, H, W), np.uint8)   # the consumer's own (touched) arrays
    def sink(r):
        if consume:
            np.copyto(got_c[r["t"]], r["bgr"]); np.copyto(got_o[r["t"]], r["output"])
        else:
            got_c[T - 1, 0, 0, 0] = r["bgr"][0, 0, 0]
    t0 = time.perf_counter()
    pipe.run(grey, bgr, sink=sink)
    dt = time.perf_counter() - t0
    out.setdefault('pipeline_host_blocked_ms_per_frame', []).append(1e3 * pipe.host_wait_s / (T - 1))
    return (T - 1) / dt, (got_c[T - 1], got_o[T - 1])


def resident():
    st = StabNetStream(params, H, W, cfg, device=dev, use_graph=True)
    d = torch.from_numpy(grey[:8]).to(dev)
    st.start(d[0:1])
    for t in range(1, 4):
        st.step(d[t:t + 1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(1, T):
        st.step(d[t % 8:t % 8 + 1])
    torch.cuda.synchronize()
    return (T - 1) / (time.perf_counter() - t0)


fs, last_s = serial()
fp, last_p = pipeline()
out["serial_fps"] = fs
out["pipeline_fps"] = fp
out["pipeline_fps_results_left_in_staging"] = pipeline(consume=False)[0]
out["resident_fps"] = resident()
out["pipeline_over_serial"] = fp / fs
out["pipeline_of_resident"] = fp / out["resident_fps"]
out["same_bytes_last_frame"] = bool(np.array_equal(last_s[0], last_p[0]) and np.array_equal(last_s[1], last_p[1]))
out["slots"] = a.slots
print(json.dumps(out))
