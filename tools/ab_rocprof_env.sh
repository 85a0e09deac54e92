#!/bin/bash
# A/B of one environment variable between two VALUES under rocprofv3 (kernel durations inside the hipGraph replay):
#   gpurun -- 'bash tools/ab_rocprof_env.sh STABNET_HEAD_PREFETCH_BLOCKS 256 0 r04_pf "warp_sample|theta_mesh"'
VAR=$1; A=$2; B=$3; TAG=$4; PAT=${5:-warp_sample}
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out
cd /tmp
for v in $A $B $A $B; do
  export $VAR=$v
  rm -rf $O/_ab_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/_ab_$v -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg --no-bf16-leg --no-roofline > $O/${TAG}_$v.json 2> $O/${TAG}_$v.err || exit 1
  cp $(find $O/_ab_$v -name "*kernel_stats.csv" | head -1) $O/${TAG}_${v}_stats.csv
  rm -rf $O/_ab_$v
  python3 -c "import json; d=json.load(open('$O/${TAG}_$v.json')); print('$VAR=$v', round(d['value'],1), 'fps', round(d['ms_per_step'],4), 'ms')"
  grep -E "$PAT" $O/${TAG}_${v}_stats.csv | awk -F, '{gsub(/"/,""); printf "    %-40s calls %s avg %.2f us\n", substr($1,1,40), $2, $4/1000}'
done
