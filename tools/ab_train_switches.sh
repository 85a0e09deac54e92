#!/bin/bash
# The training step with this round's A/B switches off (padded stem, bias column reductions, untiled interpolate backward, slab
# split of the prologue 1x1 layers) under rocprofv3, next to the shipped path:
#   gpurun -- 'bash tools/ab_train_switches.sh'   -> gpurun_out/ab_train_switches_{on,off}_kernel_stats.csv + the two bench lines
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out
cd /tmp
for name in on off; do
  if [ $name = off ]; then export STABNET_STEM_ROWRUN=0 STABNET_WGRAD_BIAS=0 STABNET_INTERP_BWD_TILED=0 STABNET_CONV_KGROUPS_PRO=0; fi
  rm -rf $O/_abs_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/_abs_$name -- python3 $R/bench.py --mode train --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $O/ab_train_switches_$name.json 2> $O/ab_train_switches_$name.err || exit 1
  cp $(find $O/_abs_$name -name "*kernel_stats.csv" | head -1) $O/ab_train_switches_${name}_kernel_stats.csv
  rm -rf $O/_abs_$name
  python3 -c "import json; d=json.loads(open('$O/ab_train_switches_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],4), 'ms/step')"
done
