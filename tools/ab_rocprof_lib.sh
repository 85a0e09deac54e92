#!/bin/bash
# A/B of two builds of the library under rocprofv3 (kernel durations inside the hipGraph replay):
#   gpurun -- 'bash tools/ab_rocprof_lib.sh <other libstabnet_hip.so> <tag> [kernel name substring]'
OTHER=$(realpath $1); TAG=$2; PAT=${3:-warp_sample}
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out
cd /tmp
for name in new old; do
  if [ $name = old ]; then export STABNET_LIB=$OTHER; else unset STABNET_LIB; fi
  rm -rf $O/_ab_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/_ab_$name -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg --no-bf16-leg --no-roofline > $O/${TAG}_$name.json 2> $O/${TAG}_$name.err || exit 1
  cp $(find $O/_ab_$name -name "*kernel_stats.csv" | head -1) $O/${TAG}_${name}_stats.csv
  rm -rf $O/_ab_$name
  python3 -c "import json; d=json.load(open('$O/${TAG}_$name.json')); print('$name', d['value'], d['ms_per_step'])"
  grep "$PAT" $O/${TAG}_${name}_stats.csv | cut -c1-60,200-
  grep "$PAT" $O/${TAG}_${name}_stats.csv | awk -F, '{print $(NF-6), $(NF-5), $(NF-4)}'
done
