#!/bin/bash
# build the probes first (on the build host; tools/bin/ is git-ignored but travels with gpurun):
#   for ab in 0 1 4 8 32 40 41 45; do hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Wno-inline-asm -Wno-unused-function \
#       -DPROBE_BF16=4 -DRING_ABLATE=$ab -I deep-online-video-stabilization_amd/csrc -o tools/bin/ring_probe_pk_$ab tools/ring_probe.hip; done
#   (-DPROBE_BF16=0 -> tools/bin/ring_probe_f32_0, -DPROBE_BF16=5 -> tools/bin/ring_probe_ps_<ablate>)
# SQ counters of the packed split kernel next to the f32-MFMA kernel (stand-alone launches of the stem and block-2 3x3 geometries)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pk_pmc; rm -rf $O; mkdir -p $O
cd /tmp
for geo in "720 1280 13 64 7 512 1" "90 160 128 128 3 512 0"; do
  set -- $geo
  tag="H$1_W$2_C$3_N$4_K$5"
  for bin in pk_0 f32_0; do
    wgs=$6; [ "$bin" = "f32_0" ] && wgs=768
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/a_${bin}_$tag -- $R/tools/bin/ring_probe_$bin $1 $2 $3 $4 $5 $wgs $7 > $O/a_${bin}_$tag.txt 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $O/b_${bin}_$tag -- $R/tools/bin/ring_probe_$bin $1 $2 $3 $4 $5 $wgs $7 > $O/b_${bin}_$tag.txt 2>&1
    date >> $O/progress.txt
  done
done
cd $R
python3 - <<'PY'
import csv, glob, os, collections
O = "gpurun_out/pk_pmc"
for d in sorted(glob.glob(O + "/[ab]_*")):
    if not os.path.isdir(d): continue
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_ring" not in r.get("Kernel_Name", ""): continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(os.path.basename(d), {k: round(v / max(n[k], 1)) for k, v in sorted(agg.items())})
PY
