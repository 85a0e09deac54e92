#!/bin/bash
# A/B of two builds of the library on the training step under rocprofv3 (per-kernel durations):
#   gpurun -- 'bash tools/ab_rocprof_train.sh <other libstabnet_hip.so> <tag> "<kernel name regex>"'
OTHER=$(realpath $1); TAG=$2; PAT=${3:-finalize}
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out
cd /tmp
for name in new old; do
  if [ $name = old ]; then export STABNET_LIB=$OTHER; else unset STABNET_LIB; fi
  rm -rf $O/_abt_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/_abt_$name -- python3 $R/bench.py --mode train --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $O/${TAG}_$name.json 2> $O/${TAG}_$name.err || exit 1
  cp $(find $O/_abt_$name -name "*kernel_stats.csv" | head -1) $O/${TAG}_${name}_stats.csv
  rm -rf $O/_abt_$name
  echo "== $name"
  python3 - "$O/${TAG}_${name}_stats.csv" "$PAT" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
for r in rows:
    if pat.search(r["Name"]):
        print("%-60s calls %5s avg %8.2f us total %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
print("sum of all kernels: %.1f us" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e3))
PY
done
