#!/bin/bash
# Fabric-side (L2 miss) read traffic (rocprofv3 --pmc FETCH_SIZE) and frame rate with and without the XCD-aware tile order.
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/xcd; rm -rf $O; mkdir -p $O
for x in 0 1; do
  export STABNET_CONV_XCD=$x
  python bench.py --no-cpu-baseline --no-train-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('XCD=$x', round(d['value'],1), 'fps')"
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f$x -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-roofline > /dev/null 2> $O/f$x.err)
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w$x -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-roofline > /dev/null 2> $O/w$x.err)
  python tools/profile_stamp.py $O/f$x $O/w$x - - $O/traffic_$x.json | head -6
done
rm -rf $O/f0 $O/f1 $O/w0 $O/w1
