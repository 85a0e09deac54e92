#!/bin/bash
# Same-box A/B of two builds of the library on the training step: gpurun -- 'bash tools/ab_train.sh <other libstabnet_hip.so> [rounds]'
OTHER=$1; ROUNDS=${2:-2}
for i in $(seq $ROUNDS); do
  for name in new old; do
    if [ $name = old ]; then export STABNET_LIB=$OTHER; else unset STABNET_LIB; fi
    python bench.py --mode train --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value'],2), round(d['ms_per_step'],4))" || exit 1
  done
done
