#!/bin/bash
# A/B of one environment switch on the same box, same build: bench.py (720p, no CPU leg) with VAR=A then VAR=B, alternated REPS times.
#   gpurun -- 'bash tools/ab_env.sh STABNET_CONV_B2B 1 0 [reps] [extra bench args]'
VAR=$1; A=$2; B=$3; REPS=${4:-2}; shift 4
for r in $(seq 1 $REPS); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train-leg --no-bf16-leg --no-roofline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$VAR=$v', round(d['value'],1), 'fps', round(d['ms_per_step'],4), 'ms', d['config'].get('launches_per_frame'), 'launches')"
  done
done
