#!/usr/bin/env python3
"""Headline benchmark: stabilised frames/s of the StabNet hot path on MI355X (BASELINE.json configs[1]).

A step = one output frame of one online stream: 13-channel stack assembled from the on-device history ring ->
ResNet-v2-50 regressor -> mesh -> multi-grid warp -> feedback push.  Frames of a stream are serially dependent
(deploy_bundle.py:322-323), so N GPUs = N independent streams (replicas, no collective): weak scaling.
Inputs (the synthetic 720p clip) are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python bench.py --gpus N --steps K --warmup W        # N > 1 without a torchrun environment: this process starts N
                                                       # fresh child ranks itself (before it touches the GPU), relays
                                                       # rank 0's JSON line and exits non-zero if fewer than N came up
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W        # the driver's form: one rank per GPU over RCCL
`n_gpus` in the line is the number of ranks the process group really has, never the flag.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec
PEAK_BF16_MFMA_TFLOPS = 2516.6   # v_mfma_f32_32x32x16_bf16: 32 cycles per 32x32x16 on each of 256 CUs x 4 SIMDs at 2.4 GHz (~2.5 PF dense)
SPLIT_PRODUCTS = 6               # bf16 x bf16 partial products per f32 product in operand mode 4 (conv_kernel.h)
PEAK_SPLIT_F32_TFLOPS = PEAK_BF16_MFMA_TFLOPS / SPLIT_PRODUCTS   # f32-equivalent matrix peak of the packed split kernels


def is_packed_kernel(name):
    return name.startswith("conv_ring_f32_kernel<") and name.split(",")[1].strip() in ("4", "5")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 200 frames; --mode train: 50 optimiser steps, SURVEY 8d)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default: 20; --mode train: 10)")
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--streams", type=int, default=1, help="streams per GPU stepped in lock-step (batch of the net)")
    ap.add_argument("--before-ch", type=int, default=31, help="accepted and ignored, as in deploy_bundle.py:15,41")
    ap.add_argument("--refine", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--clip-frames", type=int, default=32)
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer: BASELINE configs[1] (the headline); train: configs[2]/[3], siamese pairs/s")
    ap.add_argument("--train-batch", type=int, default=8, help="pairs per GPU (weak scaling)")
    ap.add_argument("--train-height", type=int, default=288)
    ap.add_argument("--train-width", type=int, default=512)
    ap.add_argument("--train-steps", type=int, default=30, help="steps of the extra train leg of the default run (5 warm-up steps)")
    ap.add_argument("--no-train-leg", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the frame eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-bf16-leg", action="store_true", help="skip the SECONDARY bf16-operand line (never the headline)")
    ap.add_argument("--operand-mode", type=int, default=4, choices=[0, 2, 3, 4],
                    help="conv operand mode of the frame (include/stabnet_hip.h, stabnet_net_set_bf16_operands): 4 = packed split "
                         "kernels (float32 operands as exact 3 x bf16 sums on the bf16 matrix pipe, f32 accumulate; f32-level "
                         "parity, the default), 0 = exact f32 MFMA")
    ap.add_argument("--no-f32-mfma-leg", action="store_true", help="skip the second figure taken with --operand-mode 0")
    ap.add_argument("--dump-event-raw", default=None, metavar="JSON",
                    help="write {kernel: raw HIP-event average} of the instrumented pass (tools/profile_stamp.py calibrates the "
                         "per-kernel event offsets against rocprofv3 with it)")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 50 if a.mode == "train" else 200
    if a.warmup is None:
        a.warmup = 10 if a.mode == "train" else 20
    return a


def csrc_sha16():
    """Hash of the kernel sources (csrc/*.hip, *.h): the PMC traffic file is only trusted for the build it was taken on."""
    import hashlib
    d = os.path.join(ROOT, "deep-online-video-stabilization_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def empirical_peaks(dev):
    """What THIS box sustains: f32 MFMA (register-only chains) and HBM copy (1 GiB, beyond the 256 MiB Infinity Cache).
    SURVEY 8d asks for the roofline fraction against the vendor peak and against the empirical one."""
    from stabnet_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    out = torch.empty(2048 * 256, dtype=torch.float32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    blocks, iters = 2048, 400
    stamps = torch.zeros(blocks * 2, dtype=torch.int64, device=dev)
    _lib.call("stabnet_probe_mfma_f32", out.data_ptr(), blocks, 50, 0, st)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        _lib.call("stabnet_probe_mfma_f32", out.data_ptr(), blocks, iters, 0, st)
    e1.record()
    torch.cuda.synchronize()
    mfma = 5 * L.stabnet_probe_mfma_f32_flops(blocks, iters) / (e0.elapsed_time(e1) * 1e-3) / 1e12
    # the clock held inside that loop (separate, stamped launch): shader cycles per 100 MHz tick, median over workgroups
    _lib.call("stabnet_probe_mfma_f32", out.data_ptr(), blocks, iters, stamps.data_ptr(), st)
    torch.cuda.synchronize()
    sp = stamps.view(blocks, 2).double()
    clock_ghz = float((sp[:, 0] / sp[:, 1].clamp(min=1)).median().item()) * 0.1
    n = 1 << 28                                             # 1 GiB of floats
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    _lib.call("stabnet_probe_hbm_copy", src.data_ptr(), dst.data_ptr(), n, st)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        _lib.call("stabnet_probe_hbm_copy", src.data_ptr(), dst.data_ptr(), n, st)
    e1.record()
    torch.cuda.synchronize()
    hbm = 5 * 8.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    torch.cuda.empty_cache()
    return {"mfma_f32_tflops": mfma, "mfma_clock_ghz": clock_ghz,
            "mfma_f32_tflops_at_2p4ghz": mfma * 2.4 / clock_ghz if clock_ghz > 0 else None, "hbm_copy_gbps": hbm,
            "how": "stabnet_probe_mfma_f32 (register-only v_mfma_f32_32x32x2_f32, 2048 WGs; clock = s_memtime / s_memrealtime "
                   "inside the loop, median over workgroups) / stabnet_probe_hbm_copy (1 GiB float4 copy, one non-temporal "
                   "float4 per thread, read+write bytes)"}


def train_leg(args, dev, dist, rank, world, steps, warmup, with_prof):
    """BASELINE configs[2]/[3]: one optimiser step = `train_batch` siamese pairs per GPU at 288x512, full forward +
    backward (incl. warp gradient) + Adam, gradient all-reduce over RCCL when world > 1.  Returns a dict."""
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    from stabnet_amd.deploy import Profiler
    from stabnet_amd.train import Trainer
    N, H, W = args.train_batch, args.train_height, args.train_width
    cfg = Config(height=H, width=W, batch_size=N)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
    pg = dist.group.WORLD if dist is not None else None
    # STABNET_FORCE_COMM=1 (one-GPU box): main() built a ONE-rank RCCL group; the buckets still go through all_reduce on the
    # communication stream, so `comm` is printed -- the launch / stream-join cost of the path, not a scaling number
    tr = Trainer(P, N, H, W, cfg, device=dev, process_group=pg, world_size=world, force_comm=pg is not None)
    b = synthetic.make_train_batch(cfg, N, H, W, seed=1234 + rank)
    dev_b = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
    gates = {"use_theta_loss": 1, "use_temp_loss": 1, "use_black_loss": 1, "use_theta_only": 0}   # late-training values
    for _ in range(warmup):
        tr.forward_backward(dev_b, gates)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if tr.comm:
        tr.comm_timing = []            # event pairs around every bucket's all-reduce on the communication stream
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.forward_backward(dev_b, gates)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    comm = None
    if tr.comm:
        # all-reduce ms per step and the fraction of it hidden under backward.  exposed = time the
        # compute stream had to wait for the collective after its own last kernel (event on the compute stream before the
        # join vs the end of the last bucket on the communication stream).
        per_step = len(tr.comm_timing) // steps
        ar_ms = sum(a.elapsed_time(b) for a, b, _ in tr.comm_timing) / steps
        exposed = sum(max(0.0, c.elapsed_time(tr.comm_timing[(i + 1) * per_step - 1][1]))
                      for i, c in enumerate(tr.compute_done)) / steps
        comm = {"allreduce_ms_per_step": ar_ms, "exposed_ms_per_step": exposed,
                "overlap_fraction": (1.0 - exposed / ar_ms) if ar_ms > 0 else None,
                "buckets_per_step": per_step, "bytes_per_step": sum(b for _, _, b in tr.comm_timing) // steps,
                "order": "reverse layer order: FC+block4, block3, block2, block1+stem, BN gamma/beta",
                "backend": dist.get_backend(), "ranks": dist.get_world_size()}
        tr.comm_timing = None
    if dist is not None:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
        dist.barrier()
    out = {"value": steps * N * world / el, "unit": "pairs/s", "ms_per_step": 1e3 * el / steps, "steps": steps,
           "warmup": warmup, "pairs_per_gpu": N, "global_batch": N * world, "height": H, "width": W,
           "tower_fwd_gflop": tr.plan.flops / 1e9, "loss": tr.losses()["total_loss"] if rank == 0 else None,
           "step_gflop_algorithmic": 6.0 * tr.plan.flops / 1e9}     # 2 towers x (fwd + dgrad + wgrad); plan.flops is per tower batch
    if comm is not None:
        out["comm"] = comm
    if with_prof:
        # the conv kernels (forward, dgrad, wgrad) record HIP-event pairs inside the library; every rank runs the same
        # instrumented steps (the collective needs all of them), rank 0 reports
        psteps = max(1, min(steps // 2, 3))
        prof = Profiler(max_records=psteps * 2000, device=dev)
        prof.calibrate()
        tab, note = load_kernel_profile((N, H, W) == (8, 288, 512), PMC_FILE_TRAIN)
        prof.set_offsets(event_offsets(tab))
        tr.prof = prof
        for _ in range(psteps):
            tr.forward_backward(dev_b, gates)
        roof, table = roofline_from_records(prof.records(), psteps)
        raw_table = roofline_from_records(prof.records(raw=True), psteps)[1]
        tr.prof = None
        if rank == 0:
            roof["whole_step_tflops"] = out["step_gflop_algorithmic"] / out["ms_per_step"]
            roof["whole_step_frac"] = roof["whole_step_tflops"] / PEAK_F32_MFMA_TFLOPS
            annotate_roofline(roof, tab, note, prof)
            out["roofline"] = roof
            out["kernels"] = table[:6]
            out["event_raw"] = {r["kernel"]: {"raw_avg_us": r["avg_us"], "launches": r["launches_per_frame"]} for r in raw_table}
    return out


PMC_FILE = os.path.join("profiles", "r04_kernel_profile_bench720p.json")
PMC_FILE_TRAIN = os.path.join("profiles", "r04_kernel_profile_train_b8.json")
PMC_FILE_1080P = os.path.join("profiles", "r04_kernel_profile_bench1080p.json")       # BASELINE configs[4] shape on one GPU
PMC_FILE_F32 = os.path.join("profiles", "r04_kernel_profile_bench720p_f32_mfma.json")  # the 720p frame with --operand-mode 0


def load_kernel_profile(workload_is_default, pmc_file=None):
    """(table, note): the stamped per-kernel profile of this workload (tools/profile_stamp.py: rocprofv3 --pmc passes, rocprofv3
    --stats averages and the HIP-event offsets calibrated against them, all taken with this same command) -- or (None, why)
    when the workload is not a BASELINE one, the file is missing, or it was taken on other kernel sources (hash of csrc/)."""
    pmc_file = pmc_file or PMC_FILE
    path = os.path.join(ROOT, pmc_file)
    if not workload_is_default:
        return None, "profiles exist for the BASELINE workloads only (720p / 1080p batch 1; training 8 pairs at 288x512)"
    if not os.path.exists(path):
        return None, "no %s (run tools/refresh_profiles.sh on the GPU box)" % pmc_file
    try:
        tab = json.load(open(path))
    except Exception as e:
        return None, "unreadable %s: %s" % (pmc_file, e)
    stamp = tab.get("__meta__", {}).get("csrc_sha16")
    if stamp != csrc_sha16():
        return None, "%s was taken on kernel sources %s, this build is %s: stale, not used" % (pmc_file, stamp, csrc_sha16())
    return tab, "%s (csrc %s)" % (pmc_file, stamp)


def profile_entry(tab, kernel):
    """The profile's entry of a kernel named as the library's profiler names it (some carry no template arguments there)."""
    if not tab:
        return None
    best = None
    for k, v in tab.items():
        if k == "__meta__":
            continue
        if k == kernel or v.get("hip_event_name") == kernel:
            return v
        if k.startswith(kernel + "<") and (best is None or v.get("rocprofv3_calls", 0) > best.get("rocprofv3_calls", 0)):
            best = v
    return best


def event_offsets(tab):
    """{library profiler kernel name: us} from the stamped profile (empty without one)."""
    out = {}
    for k, v in (tab or {}).items():
        if k != "__meta__" and "hip_event_offset_us" in v:
            out[v.get("hip_event_name", k)] = max(0.0, float(v["hip_event_offset_us"]))
    return out


def annotate_roofline(roof, tab, note, prof, algorithmic_bytes=None):
    """traffic (+ what the counter is, + the algorithmic bytes beside it), the event offset that was subtracted and rocprofv3's
    own figure for the same kernel on the same build."""
    e = profile_entry(tab, roof["kernel"])
    roof["traffic"] = e.get("l2_fabric_bytes_per_launch") if e else None
    roof["traffic_source"] = (note + ": rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes") if e else note
    roof["traffic_counts"] = ("bytes per launch on the fabric side of the eight XCD L2s (TCC -> EA requests: served by the Infinity "
                              "Cache or HBM, the counter cannot tell which) -- an upper bound of the HBM bytes")
    if algorithmic_bytes is not None:
        roof["algorithmic_bytes_per_launch"] = algorithmic_bytes
    if roof.get("traffic") and roof.get("algorithmic_bytes_per_launch"):
        roof["traffic_over_algorithmic"] = roof["traffic"] / roof["algorithmic_bytes_per_launch"]
    name = roof["kernel"]
    roof["hip_event_offset_us_subtracted"] = 1e3 * prof.offset_ms(name)
    roof["hip_event_offset_source"] = ("calibrated against the rocprofv3 average of this kernel on this build: " + note) \
        if name in prof.offsets_us else "default: half an idle event pair (%.2f us); no stamped profile for this build / workload" % (
            1e3 * prof.idle_pair_ms)
    if e and "rocprofv3_avg_us" in e:
        roof["avg_launch_us_rocprofv3"] = e["rocprofv3_avg_us"]
        work = roof.get("algorithmic_flops_per_launch") if roof["bound"] == "mfma" else roof.get("algorithmic_bytes_per_launch")
        if work:
            ach = work / (e["rocprofv3_avg_us"] * 1e-6) / (1e12 if roof["bound"] == "mfma" else 1e9)
            roof["achieved_rocprofv3"] = ach
            roof["frac_rocprofv3"] = ach / roof["peak"]
    return roof


def roofline_from_records(recs, steps):
    """recs: [(kernel, ms, flops, bytes)] over `steps` instrumented frames -> (roofline dict, per-kernel table)."""
    agg = {}
    for name, ms, fl, by in recs:
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += ms
        a[2] += fl
        a[3] += by
    table = []
    for name, (n, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        table.append({"kernel": name, "launches_per_frame": n / steps, "avg_us": 1e3 * ms / n,
                      "ms_per_frame": ms / steps, "tflops": (fl / (ms * 1e-3) / 1e12) if fl else None,
                      "gbps": by / (ms * 1e-3) / 1e9})
    dom = table[0]
    name = dom["kernel"]
    n, ms, fl, by = agg[name]
    if fl > 0:
        ach = fl / (ms * 1e-3) / 1e12
        # `achieved` counts ALGORITHMIC f32 flops.  An exact-f32-MFMA kernel is priced against the f32 MFMA peak; a packed split
        # kernel executes six bf16 MFMA flops per algorithmic flop and is priced against the bf16 dense peak / 6
        packed = is_packed_kernel(name)
        peak = PEAK_SPLIT_F32_TFLOPS if packed else PEAK_F32_MFMA_TFLOPS
        roof = {"bound": "mfma", "kernel": name, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "traffic": None, "avg_launch_us": 1e3 * ms / n,
                "algorithmic_flops_per_launch": fl / n, "algorithmic_bytes_per_launch": by / n,
                "peak_basis": ("bf16 dense MFMA peak %.1f TF / %d bf16 partial products per f32 product (operand mode 4)" % (
                    PEAK_BF16_MFMA_TFLOPS, SPLIT_PRODUCTS)) if packed else "f32 MFMA peak (v_mfma_f32_32x32x2_f32)",
                "frac_of_f32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS}
    else:
        ach = by / (ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": name, "achieved": ach, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                "frac": ach / PEAK_HBM_GBPS, "traffic": None, "avg_launch_us": 1e3 * ms / n,
                "algorithmic_bytes_per_launch": by / n}
    return roof, table


def _oracle_fps(P, clip, H, W, budget_s):
    from oracle import stabnet_oracle as O
    ocfg = O.Config(height=H, width=W)
    ring = O.DeployRing(clip[0], ocfg)
    t0 = time.time()
    n = 0
    while True:
        O.deploy_step(ring, clip[1 + n % (len(clip) - 1)], P, ocfg)
        n += 1
        el = time.time() - t0
        if el >= budget_s or el / n * (n + 1) > 1.5 * budget_s:
            break
    return n, time.time() - t0


def cpu_baseline(P, clip, H, W, budget_s):
    """The oracle (NumPy restatement of the reference, 'port') timed on this box's host cores on a bounded sample, with all
    the cores OpenBLAS takes and with ONE thread (BASELINE.md section 2 promises both)."""
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        thr = max([d.get("num_threads", 1) for d in threadpool_info()] + [1])
    except Exception:
        threadpool_limits, thr = None, os.cpu_count()
    n, el = _oracle_fps(P, clip, H, W, 0.6 * budget_s)
    out = {"value": n / el, "unit": "frames/s", "cores": thr, "host_cpus": os.cpu_count(), "kind": "port",
           "sample": "%d sequential %dx%d frames of the same synthetic clip through oracle.deploy_step "
                     "(NumPy/OpenBLAS restatement; the TF1 reference cannot run offline)" % (n, W, H)}
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            n1, el1 = _oracle_fps(P, clip, H, W, 0.4 * budget_s)
        out["single_thread"] = {"value": n1 / el1, "unit": "frames/s", "cores": 1, "sample": "%d frame(s), BLAS limited to 1 thread" % n1}
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _kfd_gpu_nodes(root="/sys/class/kfd/kfd/topology/nodes"):
    """GPU agents the kernel driver lists (nodes with SIMDs; CPU nodes have simd_count 0), read from sysfs: no HIP / HSA call."""
    n = 0
    for d in sorted(os.listdir(root)):
        with open(os.path.join(root, d, "properties")) as f:
            props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    return n


def visible_gpus(environ=None, kfd_root="/sys/class/kfd/kfd/topology/nodes"):
    """Devices this process could use, WITHOUT touching the GPU runtime: the KFD topology in sysfs, narrowed by the
    *_VISIBLE_DEVICES lists (comma-separated indices or UUIDs; an empty list hides every device).  Only when sysfs cannot be read
    does it fall back to torch.cuda.device_count(), which may open the driver -- harmless for launch_ranks(), whose rule is
    'children are started with subprocess.Popen, the parent is never replaced or re-exec'd'."""
    env = os.environ if environ is None else environ
    try:
        n = _kfd_gpu_nodes(kfd_root)
    except (OSError, ValueError):
        return torch.cuda.device_count()
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def launch_command(n, argv, env, ndev):
    """The torch.distributed.run command line that starts `n` ranks of this script, or SystemExit when the box cannot hold
    them: one rank per GPU over RCCL needs n devices (STABNET_DIST_BACKEND=gloo is the labelled rehearsal mode in which
    ranks may share a card)."""
    if env.get("STABNET_DIST_BACKEND", "nccl") == "nccl" and ndev < n:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible; one rank per GPU over RCCL needs %d "
                         "(rehearsal on fewer cards: STABNET_DIST_BACKEND=gloo)" % (n, ndev, n))
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
            "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` (N > 1) outside a torchrun environment: start N FRESH child ranks (this process has made no
    GPU call and makes none), relay their output, and return an exit code that is non-zero unless rank 0 printed a JSON
    line whose n_gpus equals N."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = launch_command(args.gpus, argv, env, visible_gpus())
    print("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    seen = None
    for ln in proc.stdout:
        sys.stdout.write(ln)
        sys.stdout.flush()
        if ln.lstrip().startswith("{") and '"metric"' in ln:
            try:
                seen = json.loads(ln)
            except ValueError:
                pass
    rc = proc.wait()
    if rc != 0:
        print("bench.py: the rank launcher exited with %d" % rc, file=sys.stderr)
        return rc
    if seen is None or seen.get("n_gpus") != args.gpus:
        print("bench.py: asked for %d ranks, the result line reports %s" % (args.gpus, None if seen is None else seen.get("n_gpus")),
              file=sys.stderr)
        return 3
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher environment has WORLD_SIZE=%d; they must agree "
                         "(n_gpus is reported from the process group)" % (args.gpus, world))
    dist = None
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)            # one rank per GPU; wraps only in the 1-GPU rehearsal below
    if world > 1 or os.environ.get("STABNET_FORCE_COMM") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        # "nccl" = RCCL over xGMI.  STABNET_DIST_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks.
        backend = os.environ.get("STABNET_DIST_BACKEND", "nccl")
        # librccl prints a five-line version banner on STDOUT when its first communicator is created; stdout carries exactly ONE
        # JSON line (the driver parses it): fd 1 points at stderr while the group and its communicator come up
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
                t0_ = torch.zeros(1, device=torch.device("cuda", dev_index))
                dist.all_reduce(t0_)                          # (the communicator is created here at the latest)
                torch.cuda.synchronize()
            else:
                dist.init_process_group(backend)
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        world = dist.get_world_size()                     # what the process group really has; this is what n_gpus reports
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    from stabnet_amd.deploy import Profiler, StabNetStream

    if args.mode == "train":
        t = train_leg(args, dev, dist, rank, world, args.steps, args.warmup, not args.no_roofline)
        if rank == 0:
            line = {"metric": "train samples/sec (siamese pairs, 288x512)", "value": t["value"], "unit": "pairs/s",
                    "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t["ms_per_step"],
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": "BASELINE.json configs[%d]: train_bundle_nobm step, %d pairs/GPU at %dx%d "
                                           "(global batch %d), two towers fwd+bwd incl. warp gradient, temporal loss, "
                                           "Adam; local BN, RCCL gradient all-reduce" % (
                                               2 if world == 1 else 3, t["pairs_per_gpu"], t["width"], t["height"],
                                               t["global_batch"]),
                               "global_batch": t["global_batch"], "parallelism": "dp%d" % world}}
            for k in ("roofline", "kernels", "loss", "tower_fwd_gflop", "comm", "step_gflop_algorithmic"):
                if k in t:
                    line[k] = t[k]
            if args.dump_event_raw and "event_raw" in t:
                json.dump(t["event_raw"], open(args.dump_event_raw, "w"), indent=1)
            print(json.dumps(line))
        if dist is not None:
            dist.destroy_process_group()
        return

    H, W, S = args.height, args.width, args.streams
    cfg = Config(height=H, width=W)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
    clip = synthetic.make_clip(H, W, args.clip_frames, seed=1234 + rank)
    clip_dev = torch.from_numpy(clip).to(dev)                      # resident in HBM before timing
    frames = [clip_dev[t:t + 1].expand(S, H, W).contiguous() for t in range(args.clip_frames)]
    stream = StabNetStream(P, H, W, cfg, streams=S, device=dev, refine=args.refine, before_ch=args.before_ch,
                           use_graph=not args.no_graph, operand_mode=args.operand_mode)
    stream.start(frames[0])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    t = 1
    for _ in range(args.warmup):
        stream.step(frames[t % args.clip_frames])
        t += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stream.step(frames[t % args.clip_frames])
        t += 1
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    barrier()
    checksum = float(stream.out_img.double().sum().item())
    from stabnet_amd import _lib as _sl
    plan_flops, plan_launches = stream.reg.plan.flops, stream.reg.plan.num_launches
    frame_launches = _sl.lib().stabnet_deploy_frame_launches(stream.reg.plan.handle, cfg.grid_h, cfg.grid_w)
    graph_used = stream._graph is not None

    roof, table, prof_ms, roof_warp = None, None, None, None
    if rank == 0 and not args.no_roofline:
        # the same K steps again with an event pair around every launch (instrumentation kept out of `value`)
        prof = Profiler(max_records=args.steps * (stream.reg.plan.num_launches + 16))
        prof.calibrate()
        # PMC / rocprofv3 profiles exist for the two BASELINE inference shapes (720p = configs[1], 1080p = the per-GPU shape of configs[4])
        pmc_file = {(720, 1280): PMC_FILE, (1080, 1920): PMC_FILE_1080P}.get((H, W))
        if args.operand_mode == 0:                           # the f32-MFMA frame has its own stamped profile (720p only)
            pmc_file = PMC_FILE_F32 if (H, W) == (720, 1280) else None
        pmc_ok = pmc_file is not None and (S, args.refine) == (1, 1) and args.operand_mode in (0, 4)
        ktab, knote = load_kernel_profile(pmc_ok, pmc_file)
        prof.set_offsets(event_offsets(ktab))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            stream.step(frames[t % args.clip_frames], prof)
            t += 1
        torch.cuda.synchronize()
        prof_ms = 1e3 * (time.perf_counter() - t1) / args.steps
        roof, table = roofline_from_records(prof.records(), args.steps)
        raw_table = roofline_from_records(prof.records(raw=True), args.steps)[1]
        if args.dump_event_raw:
            json.dump({r["kernel"]: {"raw_avg_us": r["avg_us"], "launches": r["launches_per_frame"]} for r in raw_table},
                      open(args.dump_event_raw, "w"), indent=1)
        for row in table:                      # the HBM-bound kernel of the path: the fused map + gather warp
            if row["kernel"] == "warp_sample_kernel":
                # sampler + feedback push in one launch: 20 HW (src, out, black, maps) + 12 HW (ring frame, ring mask, frame_fb)
                roof_warp = {"bound": "hbm", "kernel": "warp_sample_kernel", "achieved": row["gbps"], "peak": PEAK_HBM_GBPS,
                             "unit": "GB/s", "frac": row["gbps"] / PEAK_HBM_GBPS, "avg_launch_us": row["avg_us"],
                             "algorithmic_bytes_per_launch": S * (32.0 * H * W + 776.0)}
                annotate_roofline(roof_warp, ktab, knote, prof)
                roof_warp["kernel"] = "warp_sample_kernel<1>"
        annotate_roofline(roof, ktab, knote, prof)

    f32leg = None
    if rank == 0 and args.operand_mode != 0 and not args.no_f32_mfma_leg:
        # the same frames on the exact f32 MFMA kernels (operand mode 0: the headline of rounds 1-3), and what separates the two
        # modes on ONE frame from identical ring state
        s0 = StabNetStream(P, H, W, cfg, streams=S, device=dev, refine=args.refine, use_graph=not args.no_graph, operand_mode=0)
        s0.start(frames[0])
        s0.frames_ring.copy_(stream.frames_ring); s0.masks_ring.copy_(stream.masks_ring); s0.head_dev.copy_(stream.head_dev)
        tt = t
        for _ in range(max(3, args.warmup // 2)):
            s0.step(frames[tt % args.clip_frames]); tt += 1
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            s0.step(frames[tt % args.clip_frames]); tt += 1
        torch.cuda.synchronize()
        el0 = time.perf_counter() - t2
        s0.frames_ring.copy_(stream.frames_ring); s0.masks_ring.copy_(stream.masks_ring); s0.head_dev.copy_(stream.head_dev)
        a = stream.step(frames[t % args.clip_frames])["theta"].clone()
        stream.frames_ring.copy_(s0.frames_ring); stream.masks_ring.copy_(s0.masks_ring); stream.head_dev.copy_(s0.head_dev)
        b = s0.step(frames[t % args.clip_frames])["theta"].clone()
        f32leg = {"label": "the same frames with --operand-mode 0: exact f32 MFMA kernels (v_mfma_f32_32x32x2_f32)",
                  "value": args.steps * S / el0, "unit": "frames/s", "ms_per_step": 1e3 * el0 / args.steps,
                  "whole_frame_tflops": plan_flops / (el0 / args.steps) / 1e12,
                  "whole_frame_frac_of_f32_mfma_peak": plan_flops / (el0 / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                  "theta_max_abs_dev_headline_vs_this": float((a - b).abs().max().item()), "theta_scale": float(b.abs().max().item())}
        del s0
    bf16 = None
    if rank == 0 and not args.no_bf16_leg:
        # SECONDARY, labelled: the same frames with the conv operands rounded to bf16 at fragment-read time (fp32 tensors,
        # fp32 accumulate).  Reported beside the fp32 headline with the deviation it costs; `value` stays the fp32 figure.
        th32 = stream.theta.clone()
        s16 = StabNetStream(P, H, W, cfg, streams=S, device=dev, refine=args.refine, use_graph=not args.no_graph,
                            bf16_operands=1)
        s16.start(frames[0])
        s16.frames_ring.copy_(stream.frames_ring); s16.masks_ring.copy_(stream.masks_ring); s16.head_dev.copy_(stream.head_dev)
        tt = t
        for _ in range(max(3, args.warmup // 2)):
            s16.step(frames[tt % args.clip_frames]); tt += 1
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            s16.step(frames[tt % args.clip_frames]); tt += 1
        torch.cuda.synchronize()
        el16 = time.perf_counter() - t2
        # deviation on ONE frame from identical ring state
        s16.frames_ring.copy_(stream.frames_ring); s16.masks_ring.copy_(stream.masks_ring); s16.head_dev.copy_(stream.head_dev)
        a = stream.step(frames[t % args.clip_frames])["theta"].clone()
        stream.frames_ring.copy_(s16.frames_ring); stream.masks_ring.copy_(s16.masks_ring); stream.head_dev.copy_(s16.head_dev)
        b = s16.step(frames[t % args.clip_frames])["theta"].clone()
        bf16 = {"label": "SECONDARY: bf16 conv operands (rounded at fragment-read time), fp32 tensors + fp32 accumulate; "
                         "not the reference's precision, never the headline",
                "value": args.steps * S / el16, "unit": "frames/s", "ms_per_step": 1e3 * el16 / args.steps,
                "theta_max_abs_dev_vs_headline": float((a - b).abs().max().item()), "theta_scale": float(a.abs().max().item())}
        del s16, th32
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(P, clip, H, W, args.cpu_baseline_seconds)
    train = None
    if not args.no_train_leg:
        del stream
        torch.cuda.empty_cache()
        train = train_leg(args, dev, dist, rank, world, args.train_steps, 5, not args.no_roofline)
    peaks = empirical_peaks(dev) if (rank == 0 and not args.no_roofline) else None

    if dist is not None:
        dist.barrier()
    if rank == 0:
        fps = args.steps * S * world / el
        line = {
            "metric": "stabilized frames/sec (720p, before_ch=31)" if (H, W) == (720, 1280) else
                      "stabilized frames/sec (%dx%d)" % (W, H),
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # operands, results and accumulation are float32 in every headline mode; mode 4 multiplies them as exact 3 x bf16 splits
            "dtype": "f32" if args.operand_mode == 0 else "f32 (bf16x3 split MFMA, f32 accumulate)",
            "arithmetic": "exact f32 MFMA (v_mfma_f32_32x32x2_f32)" if args.operand_mode == 0 else
                          "every f32 conv operand is the exact sum of 3 bf16 terms; %d bf16 x bf16 partial products per f32 product on "
                          "v_mfma_f32_32x32x16_bf16, f32 accumulate; f32-level results (theta 1e-7 from the oracle, error against a float64 "
                          "convolution = the f32 MFMA kernels': tests/test_operand_mode4_gpu.py, tests/test_conv_packed_gpu.py)" % (
                              9 if args.operand_mode == 3 else 6),
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: v2_93 net, %dx%d, %d stream(s)/GPU, batch=1 per stream, "
                                   "13-ch stack from a 32-deep ring (lags 1,2,4,8,16,32; --before-ch %d ignored as in "
                                   "the reference), ResNet-v2-50 regressor + 4x4 multi-grid warp + feedback; one "
                                   "independent stream set per GPU (replicas only)" % (W, H, S, args.before_ch),
                       "height": H, "width": W, "streams_per_gpu": S, "refine": args.refine,
                       "backbone_gflop_per_frame": plan_flops / 1e9 / S,
                       "operand_mode": args.operand_mode,
                       "launches_per_frame": frame_launches + 1, "hip_graph": bool(graph_used)},   # + the frame copy into the graph's input
            "per_gpu_fps": fps / world, "checksum": checksum,
            # BASELINE.json words the metric per GPU; `value` is the whole-job aggregate the bench contract asks for and
            # `per_gpu_fps` the per-GPU figure (identical at N = 1); the training half of the metric is the `train` object
            "baseline_metric": "stabilized frames/sec/GPU (720p, before_ch=31) + train samples/sec at 1/2/4/8 GPU",
        }
        if roof is not None:
            line["roofline"] = roof
            if roof_warp is not None:
                line["roofline_warp"] = roof_warp
            line["kernels"] = table[:8]
            line["instrumented_ms_per_step"] = prof_ms
            line["roofline"]["idle_event_pair_us"] = 1e3 * prof.idle_pair_ms
            line["roofline"]["whole_frame_tflops"] = plan_flops / (el / args.steps) / 1e12
            # the frame's algorithmic f32 flops per second against the f32 MFMA peak (comparable across rounds and modes; a packed
            # split frame may exceed what f32 MFMA instructions could do) and, in mode 4, against the split kernels' own peak
            line["roofline"]["whole_frame_frac"] = line["roofline"]["whole_frame_tflops"] / PEAK_F32_MFMA_TFLOPS
            if args.operand_mode == 4:
                line["roofline"]["whole_frame_frac_of_split_peak"] = line["roofline"]["whole_frame_tflops"] / PEAK_SPLIT_F32_TFLOPS
        if peaks is not None:
            line["empirical_peaks"] = peaks
            if roof is not None and roof.get("bound") == "mfma" and not is_packed_kernel(roof["kernel"]):   # (the probe is an f32 MFMA chain)
                line["roofline"]["frac_of_empirical_peak"] = roof["achieved"] / peaks["mfma_f32_tflops"]
            if roof_warp is not None:
                line["roofline_warp"]["frac_of_empirical_peak"] = roof_warp["achieved"] / peaks["hbm_copy_gbps"]
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if f32leg is not None:
            line["f32_mfma_mode"] = f32leg
        if bf16 is not None:
            line["secondary_bf16_operands"] = bf16
        if train is not None:
            train.pop("event_raw", None)
            line["train"] = train          # BASELINE configs[2]/[3] measured in the same run (second metric)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
