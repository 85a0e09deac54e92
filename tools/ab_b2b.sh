#!/bin/bash
# in-network A/B of the conv2 -> conv3 fusion rule (720p): off / the default window / every eligible unit
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train-leg --no-bf16-leg --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$*', round(d['value'],1), 'fps', round(d['ms_per_step'],4), 'ms', d['config'].get('launches_per_frame'), 'launches')"; }
for r in 1 2; do
  run STABNET_CONV_B2B_PLAN=0
  run STABNET_CONV_B2B_PLAN=1 STABNET_CONV_B2B_MAX_TILES=256 STABNET_CONV_B2B_CMASK=1
  run STABNET_CONV_B2B_PLAN=1
  run STABNET_CONV_B2B_PLAN=1 STABNET_CONV_B2B_MAX_TILES=256
done
