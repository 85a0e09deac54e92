#!/bin/bash
# Regenerates the judged artefacts under profiles/ on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r03'
# Everything lands in gpurun_out/refresh/ and, named per round, in gpurun_out/profiles_<round>/ ready to copy to profiles/.
set -o pipefail
R=$PWD
ROUND=${1:-r03}
export TMPDIR=/tmp
O=$R/gpurun_out/refresh; rm -rf $O; mkdir -p $O
P=$R/gpurun_out/profiles_$ROUND; rm -rf $P; mkdir -p $P
cd /tmp
# PMC first: bench.py reads the traffic file (stamped with the kernel-source hash) when it prints `roofline.traffic`
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-roofline > /dev/null 2> $O/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-roofline > /dev/null 2> $O/pmc_write.err || exit 1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $P/${ROUND}_pmc_hbm_traffic_bench720p.json || exit 1
mkdir -p profiles && cp $P/${ROUND}_pmc_hbm_traffic_bench720p.json profiles/     # (on the box: so that the next command sees it)
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_train -- python3 $R/bench.py --mode train --steps 3 --warmup 2 --no-roofline > /dev/null 2> $O/pmc_fetch_train.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_train -- python3 $R/bench.py --mode train --steps 3 --warmup 2 --no-roofline > /dev/null 2> $O/pmc_write_train.err || exit 1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch_train $O/pmc_write_train $P/${ROUND}_pmc_hbm_traffic_train_b8.json "bench.py --mode train --steps 3 --warmup 2 (8 pairs, 288x512)" || exit 1
cp $P/${ROUND}_pmc_hbm_traffic_train_b8.json profiles/
python bench.py > $P/${ROUND}_bench_720p.json 2> $O/bench.err || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_infer -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg --no-bf16-leg > $O/stats_infer.json 2> $O/stats_infer.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -- python3 $R/bench.py --mode train --steps 10 --warmup 3 --no-roofline > $O/stats_train.json 2> $O/stats_train.err || exit 1
cd $R
cp $(find $O/stats_infer -name "*kernel_stats.csv" | head -1) $P/${ROUND}_rocprofv3_kernel_stats_bench720p.csv
cp $(find $O/stats_train -name "*kernel_stats.csv" | head -1) $P/${ROUND}_rocprofv3_kernel_stats_train_b8.csv
python bench.py --mode train --steps 20 --warmup 5 > $P/${ROUND}_bench_train_1gpu.json 2> $O/train.err || exit 1
python tools/layer_table.py > $P/${ROUND}_layers_720p.txt 2> $O/layers.err || true
python tools/layer_table.py --bf16 > $P/${ROUND}_layers_720p_bf16_operands.txt 2> $O/layers_bf16.err || true
# BASELINE configs[4] shape on one GPU (1080p, batch 1): PMC passes first (roofline_warp.traffic of the 1080p line reads them)
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_1080p -- python3 $R/bench.py --height 1080 --width 1920 --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-roofline --no-bf16-leg > /dev/null 2> $O/pmc_fetch_1080p.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_1080p -- python3 $R/bench.py --height 1080 --width 1920 --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-roofline --no-bf16-leg > /dev/null 2> $O/pmc_write_1080p.err || exit 1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch_1080p $O/pmc_write_1080p $P/${ROUND}_pmc_hbm_traffic_bench1080p.json "bench.py --height 1080 --width 1920 --steps 20 --warmup 5 (1080p, batch 1)" || exit 1
cp $P/${ROUND}_pmc_hbm_traffic_bench1080p.json profiles/
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_1080p -- python3 $R/bench.py --height 1080 --width 1920 --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg --no-bf16-leg > $P/${ROUND}_bench_1080p.json 2> $O/stats_1080p.err || exit 1
cd $R
cp $(find $O/stats_1080p -name "*kernel_stats.csv" | head -1) $P/${ROUND}_rocprofv3_kernel_stats_bench1080p.csv
bash tools/capacity_sweep.sh > $P/${ROUND}_capacity_sweep.txt 2> $O/capacity.err || true
ls -la $P
