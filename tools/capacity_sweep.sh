#!/bin/bash
# What the 288 GB of one MI355X buy beyond the BASELINE configurations (which were sized for 12 GB cards): more pairs per GPU in
# training and more lock-stepped 720p streams per GPU in serving.  Run through gpurun from the repo root; prints one line per point.
set -o pipefail
for b in 8 16 32 64; do
  timeout -k 10 300 python bench.py --mode train --train-batch $b --steps 10 --warmup 3 --no-roofline 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train pairs/GPU=%d: %.1f pairs/s, %.2f ms/step' % ($b, d['value'], d['ms_per_step']))" || echo "train pairs/GPU=$b failed"
done
for s in 1 2 4 8 16; do
  timeout -k 10 300 python bench.py --streams $s --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg --no-roofline --no-bf16-leg 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('720p streams/GPU=%d: %.1f frames/s, %.2f ms per lock-step' % ($s, d['value'], d['ms_per_step']))" || echo "streams=$s failed"
done
