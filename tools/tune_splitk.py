#!/usr/bin/env python3
"""Measure the split-K of every convolution IN THE NETWORK (per-launch HIP-event timings of whole frames / training steps,
kernel + its reduce launch + the average launch gap) and write csrc/conv_tuning_table.h.
  python tools/tune_splitk.py [--out deep-online-video-stabilization_amd/csrc/conv_tuning_table.h]
Workloads: deploy 1280x720 batch 1 (BASELINE configs[1]) and training 8 pairs at 288x512 (configs[2])."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabnet_amd import _lib, synthetic
from stabnet_amd.config import Config
from stabnet_amd.deploy import Profiler, StabNetStream

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                              "deep-online-video-stabilization_amd", "csrc", "conv_tuning_table.h"))
ap.add_argument("--reps", type=int, default=16)
ap.add_argument("--gap-us", type=float, default=2.6, help="average launch gap charged to every extra (reduce) launch")
ap.add_argument("--skip-train", action="store_true")
ap.add_argument("--skip-infer", action="store_true")
ap.add_argument("--infer-sizes", default="720x1280", help="comma list of HxW deploy workloads, e.g. 720x1280,1080x1920,288x512,256x256")
ap.add_argument("--mode", type=int, default=0, help="conv operand mode of the deploy workloads (4: packed split kernels -> conv_tuning_table_packed.h)")
ap.add_argument("--merge", action="store_true", help="keep the entries of the existing table for shapes not measured in this run")
a = ap.parse_args()
L = _lib.lib()
CANDS = [1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20]


def key_of(name, shp):
    M, N, K, _ = shp
    if name.startswith("conv_ring_f32_kernel<0, 0, 1, 1>") or name.startswith("conv_ring_f32_kernel<0, 0, 2, 1>") or name.startswith("conv_ring_f32_kernel<0, 4, 1, 1>") or name.startswith("conv_ring_f32_kernel<0, 4, 2, 1>"):   # the fragment-prologue forms run the register-staged kernel's plans
        return (M, N, K, 1, 0)
    if name.startswith("conv_ring_f32_kernel<0"):
        return (M, N, K, 1, 1)
    if name.startswith("conv_ring_f32_kernel<1"):
        return (M, N, K, 3, 1)
    if name.startswith("conv_ring_f32_kernel<2"):
        return (M, N, K, 7, 1)
    mode = int(name.split(",")[5].strip(" >"))          # conv_igemm_f32_kernel<BM, BN, BK, WM, WN, MODE, ...> and ..._pair_kernel<..., MODE>
    kh = 1 if mode == 0 else (7 if K == 784 else 3)
    return (M, N, K, kh, 0)


def conv_layers(recs, per):
    """per-launch medians over the reps -> [(key, us incl. reduce + gap)] in launch order."""
    reps = len(recs) // per
    out = []
    j = 0
    while j < per:
        rs = [recs[r * per + j] for r in range(reps)]
        name, shp = rs[0][0], rs[0][4]
        us = 1e3 * float(np.median([r[1] for r in rs]))
        if name.startswith("conv_ring") or name.startswith("conv_igemm"):
            if shp[3] > 1 and j + 1 < per and recs[j + 1][0].startswith("conv_splitk_reduce"):
                us += 1e3 * float(np.median([recs[r * per + j + 1][1] for r in range(reps)])) + a.gap_us
                j += 1
            out.append((key_of(name, shp), us))
        j += 1
    return out


def set_table(table):
    L.stabnet_conv_tuning_table_set(-1, 0, 0, 0, 0, 0)
    for (M, N, K, kh, ring), s in table.items():
        L.stabnet_conv_tuning_table_set(M, N, K, kh, ring, s)


INFER_HW = (720, 1280)


def run_infer(table):
    set_table(table)
    H, W = INFER_HW
    cfg = Config(height=H, width=W)
    P = synthetic.make_params(cfg, 0, 0.2)
    clip = torch.from_numpy(synthetic.make_clip(H, W, 4, 1234)).cuda()
    s = StabNetStream(P, H, W, cfg, streams=1, operand_mode=a.mode)
    fr = [clip[t:t + 1].contiguous() for t in range(4)]
    s.start(fr[0])
    for i in range(5):
        s.step(fr[i % 4])
    prof = Profiler(a.reps * 200)
    prof.calibrate()
    for i in range(a.reps):
        s.step(fr[i % 4], prof)
    recs = prof.records_with_shapes()
    per = len(recs) // a.reps
    return conv_layers(recs, per)


def run_train(table):
    set_table(table)
    from stabnet_amd.train import Trainer
    N, H, W = 8, 288, 512
    cfg = Config(height=H, width=W, batch_size=N)
    tr = Trainer(synthetic.make_params(cfg, 0, 0.2), N, H, W, cfg, device=torch.device("cuda:0"))
    b = {k: torch.from_numpy(v).cuda() for k, v in synthetic.make_train_batch(cfg, N, H, W, 5).items()}
    gates = {"use_theta_loss": 1, "use_temp_loss": 1, "use_black_loss": 1, "use_theta_only": 0}
    for _ in range(2):
        tr.forward_backward(b, gates)
    reps = max(3, a.reps // 4)
    prof = Profiler(reps * 1500)
    prof.calibrate()
    tr.prof = prof
    for _ in range(reps):
        tr.forward_backward(b, gates)
    torch.cuda.synchronize()
    tr.prof = None
    recs = prof.records_with_shapes()
    per = len(recs) // reps
    out = conv_layers(recs, per)
    del tr
    torch.cuda.empty_cache()
    return out


def tune(run, label):
    base = run({})
    keys = sorted(set(k for k, _ in base))
    tot = {k: {} for k in keys}
    for c in CANDS:
        layers = run({k: c for k in keys})
        for k, us in layers:
            tot[k][c] = tot[k].get(c, 0.0) + us
    base_tot = {}
    for k, us in base:
        base_tot[k] = base_tot.get(k, 0.0) + us
    best = {}
    print("# %s: shape key (M, Cout, K, KH, ring): default-rule us -> best split (us)" % label)
    for k in keys:
        fastest = min(tot[k].values())
        c = min(cc for cc in tot[k] if tot[k][cc] <= 1.01 * fastest + 0.3)       # smallest split within 1 % of the fastest
        best[k] = c
        print("#   %-34s %8.1f -> s=%-2d %8.1f   %s" % (k, base_tot[k], c, tot[k][c],
                                                         " ".join("%d:%.0f" % (cc, tot[k][cc]) for cc in CANDS)))
    final = run(best)
    print("# %s: conv time per step: rule %.1f us -> table %.1f us" % (label, sum(u for _, u in base), sum(u for _, u in final)))
    return best


table = {}
if a.merge and os.path.exists(a.out):
    import re
    for m in re.finditer(r"\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\},", open(a.out).read()):
        v = [int(x) for x in m.groups()]
        if v[0] > 0:
            table[tuple(v[:5])] = v[5]
for hw in ([] if a.skip_infer else a.infer_sizes.split(",")):
    INFER_HW = tuple(int(v) for v in hw.split("x"))
    for k, v in tune(run_infer, "deploy %dx%d batch 1" % (INFER_HW[1], INFER_HW[0])).items():
        table[k] = v
if not a.skip_train:
    t2 = tune(run_train, "train 8 x 288x512")
    for k, v in t2.items():
        if a.skip_infer:
            table[k] = v                 # a training-only run re-measures the training shapes: its results replace the merged entries
        else:
            table.setdefault(k, v)
L.stabnet_conv_tuning_table_set(-1, 0, 0, 0, 0, 0)
with open(a.out, "w") as f:
    if a.mode == 4:
        f.write("// Measured split-K choices {M, Cout, K, KH, ring, splitk} for plans that run the packed split kernels (operand mode 4:\n")
        f.write("// conv_ring_f32_kernel<MODE, 4, KG, PRO>, two workgroups per CU, a two-way split inside the workgroup); GENERATED by\n")
        f.write("// tools/tune_splitk.py --mode 4 on MI355X -- do not edit.  Shapes not listed fall back to conv_tuning_table.h.\n")
        f.write("// Per shape: the smallest split within 1 % of the fastest measured one (kernel + reduce launch + launch gap, in-network).\n")
        f.write("static const TuneEntry g_tuning_packed[] = {\n")
    else:
        f.write("// Measured split-K choices {M, Cout, K, KH, ring, splitk}; GENERATED by tools/tune_splitk.py on MI355X -- do not edit.\n")
        f.write("// Workloads: deploy batch 1 at 1280x720, 1920x1080, 512x288 and 256x256 (BASELINE configs[1], [4], the reference's native\n")
        f.write("// size, configs[0]); training 8 pairs at 288x512 (forward with BN prologue: ring 0, both towers as one launch where the\n")
        f.write("// tile allows it, i.e. M of 16 samples; dgrad of the pair: ring 1).\n")
        f.write("// Per shape: the smallest split within 1 % of the fastest measured one (kernel + reduce launch + launch gap, in-network).\n")
        f.write("static const TuneEntry g_tuning_builtin[] = {\n")
    for (M, N, K, kh, ring), s in sorted(table.items()):
        f.write("    {%d, %d, %d, %d, %d, %d},\n" % (M, N, K, kh, ring, s))
    f.write("    {0, 0, 0, 0, 0, 0},\n};\n")
print("wrote", a.out, len(table), "entries")
