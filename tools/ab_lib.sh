#!/bin/bash
# A/B of two builds of the library on the same box (no profiler): bench.py at 720p, alternated REPS times.
#   gpurun -- 'bash tools/ab_lib.sh <other libstabnet_hip.so> [reps] [extra bench args]'
OTHER=$(realpath $1); REPS=${2:-2}; shift 2
for r in $(seq 1 $REPS); do
  for name in new old; do
    if [ $name = old ]; then export STABNET_LIB=$OTHER; else unset STABNET_LIB; fi
    python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-bf16-leg --no-roofline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); t=d.get('train') or {}; print('$name', round(d['value'],1), d['unit'], round(d['ms_per_step'],4), 'ms', d['config'].get('launches_per_frame'), 'launches | train', t.get('value'))"
  done
done
