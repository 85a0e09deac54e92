#!/bin/bash
# Regenerates the judged artefacts under profiles/ on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r04'
# Everything lands in gpurun_out/refresh/ and, named per round, in gpurun_out/profiles_<round>/ ready to copy to profiles/.
#
# Per workload (720p inference, 8-pair training step, 1080p inference) the SAME bench command runs four times: two rocprofv3 --pmc
# passes (FETCH_SIZE, WRITE_SIZE: separate runs), one rocprofv3 --kernel-trace --stats run, one plain run that dumps the raw
# HIP-event average per kernel.  tools/profile_stamp.py merges them into <round>_kernel_profile_<workload>.json, stamped with the hash
# of csrc/: bench.py then subtracts the per-kernel event offset (raw event - rocprofv3) from its live event times and reports the
# traffic next to the algorithmic bytes.  The headline bench line is taken LAST, with the stamped files in place.
set -o pipefail
R=$PWD
ROUND=${1:-r04}
export TMPDIR=/tmp
O=$R/gpurun_out/refresh; rm -rf $O; mkdir -p $O
( while true; do date >> $O/heartbeat.txt; sleep 60; done ) &      # (a counter pass at 1080p is silent for minutes: gpurun takes 7 silent minutes for a hang)
HEARTBEAT=$!
trap "kill $HEARTBEAT 2>/dev/null" EXIT
P=$R/gpurun_out/profiles_$ROUND; rm -rf $P; mkdir -p $P
mkdir -p $R/profiles

workload() {   # tag, stats csv name, out json name, command text, bench args...
  local TAG=$1 CSV=$2 OUT=$3 TXT=$4; shift 4
  cd /tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$TAG -- python3 $R/bench.py "$@" --no-roofline > /dev/null 2> $O/pmc_fetch_$TAG.err || return 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$TAG -- python3 $R/bench.py "$@" --no-roofline > /dev/null 2> $O/pmc_write_$TAG.err || return 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$TAG -- python3 $R/bench.py "$@" > $O/stats_$TAG.json 2> $O/stats_$TAG.err || return 1
  cd $R
  cp $(find $O/stats_$TAG -name "*kernel_stats.csv" | head -1) $P/$CSV || return 1
  python bench.py "$@" --dump-event-raw $O/event_raw_$TAG.json > /dev/null 2> $O/event_raw_$TAG.err || return 1
  python tools/profile_stamp.py $O/pmc_fetch_$TAG $O/pmc_write_$TAG $P/$CSV $O/event_raw_$TAG.json $P/$OUT "$TXT" || return 1
  cp $P/$OUT profiles/                         # (on the box: so that the bench runs below see it)
}

workload 720p ${ROUND}_rocprofv3_kernel_stats_bench720p.csv ${ROUND}_kernel_profile_bench720p.json \
  "bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg --no-bf16-leg --no-f32-mfma-leg (720p, batch 1, operand mode 4)" \
  --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg --no-bf16-leg --no-f32-mfma-leg || exit 1
workload 720p_f32 ${ROUND}_rocprofv3_kernel_stats_bench720p_f32_mfma.csv ${ROUND}_kernel_profile_bench720p_f32_mfma.json \
  "bench.py --operand-mode 0 --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg --no-bf16-leg (720p, batch 1, exact f32 MFMA kernels)" \
  --operand-mode 0 --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg --no-bf16-leg || exit 1
workload train ${ROUND}_rocprofv3_kernel_stats_train_b8.csv ${ROUND}_kernel_profile_train_b8.json \
  "bench.py --mode train --steps 10 --warmup 3 (8 pairs, 288x512)" \
  --mode train --steps 10 --warmup 3 || exit 1
workload 1080p ${ROUND}_rocprofv3_kernel_stats_bench1080p.csv ${ROUND}_kernel_profile_bench1080p.json \
  "bench.py --height 1080 --width 1920 --steps 50 --warmup 5 --no-cpu-baseline --no-train-leg --no-bf16-leg --no-f32-mfma-leg (1080p, batch 1, operand mode 4)" \
  --height 1080 --width 1920 --steps 50 --warmup 5 --no-cpu-baseline --no-train-leg --no-bf16-leg --no-f32-mfma-leg || exit 1

cd $R
python bench.py > $P/${ROUND}_bench_720p.json 2> $O/bench.err || exit 1
python bench.py --operand-mode 0 --no-cpu-baseline --no-train-leg --no-bf16-leg > $P/${ROUND}_bench_720p_f32_mfma.json 2> $O/bench_f32.err || exit 1
python bench.py --mode train > $P/${ROUND}_bench_train_1gpu.json 2> $O/train.err || exit 1
python bench.py --height 1080 --width 1920 --no-cpu-baseline --no-train-leg --no-bf16-leg > $P/${ROUND}_bench_1080p.json 2> $O/bench1080.err || exit 1
# multi-GPU readiness on ONE GPU: the same training step through a one-rank RCCL group (5 collectives + stream joins per step), next to
# the no-group run above on the same box
STABNET_FORCE_COMM=1 python bench.py --mode train --no-roofline 2> $O/train_rccl.err | grep '^{' > $P/${ROUND}_bench_train_1rank_rccl.json || true
python tools/layer_table.py --mode 4 > $P/${ROUND}_layers_720p.txt 2> $O/layers.err || true                      # the default operand mode of bench.py
python tools/layer_table.py --mode 0 > $P/${ROUND}_layers_720p_f32_mfma.txt 2> $O/layers_f32.err || true
python tools/layer_table.py --bf16 > $P/${ROUND}_layers_720p_bf16_operands.txt 2> $O/layers_bf16.err || true
python tools/check_pinned_regs.py > $P/${ROUND}_pinned_regs_check.txt 2>&1 || true
bash tools/capacity_sweep.sh > $P/${ROUND}_capacity_sweep.txt 2> $O/capacity.err || true
ls -la $P
