#!/bin/bash
# Regenerates the judged artefacts under profiles/ on the GPU box (run through gpurun from the repo root).
set -o pipefail
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/refresh; rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_infer -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg > $O/stats_infer.json 2> $O/stats_infer.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -- python $R/bench.py --mode train --steps 10 --warmup 3 --no-roofline > $O/stats_train.json 2> $O/stats_train.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-roofline > /dev/null 2> $O/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-roofline > /dev/null 2> $O/pmc_write.err || exit 1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json
find $O -name "*kernel_stats.csv" | head
