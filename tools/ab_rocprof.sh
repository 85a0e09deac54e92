#!/bin/bash
# A/B of one environment switch under rocprofv3 (kernel durations inside the hipGraph replay):
#   gpurun -- 'bash tools/ab_rocprof.sh STABNET_CONV_KGROUPS r03_kg'
# writes gpurun_out/<tag>_{on,off}_stats.csv and prints the two frame rates.
VAR=$1; TAG=$2
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out
cd /tmp
for v in 1 0; do
  name=$([ $v = 1 ] && echo on || echo off)
  export $VAR=$v
  rm -rf $O/_ab_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/_ab_$name -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg --no-bf16-leg --no-roofline > $O/${TAG}_$name.json 2> $O/${TAG}_$name.err || exit 1
  cp $(find $O/_ab_$name -name "*kernel_stats.csv" | head -1) $O/${TAG}_${name}_stats.csv
  rm -rf $O/_ab_$name
  python3 -c "import json; d=json.load(open('$O/${TAG}_$name.json')); print('$VAR=$v', d['value'], d['ms_per_step'], d['config']['launches_per_frame'])"
done
