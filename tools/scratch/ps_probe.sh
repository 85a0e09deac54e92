#!/bin/bash
# build the probes first (on the build host; tools/bin/ is git-ignored but travels with gpurun):
#   for ab in 0 1 4 8 32 40 41 45; do hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Wno-inline-asm -Wno-unused-function \
#       -DPROBE_BF16=4 -DRING_ABLATE=$ab -I deep-online-video-stabilization_amd/csrc -o tools/bin/ring_probe_pk_$ab tools/ring_probe.hip; done
#   (-DPROBE_BF16=0 -> tools/bin/ring_probe_f32_0, -DPROBE_BF16=5 -> tools/bin/ring_probe_ps_<ablate>)
for geo in "90 160 128 128 3" "45 80 512 1280 1" "720 1280 13 64 7"; do
  set -- $geo
  stem=0; [ "$5" = "7" ] && stem=1
  echo "== H=$1 W=$2 Cin=$3 N=$4 KH=$5"
  tools/bin/ring_probe_pk_0 $1 $2 $3 $4 $5 512 $stem
  tools/bin/ring_probe_ps_0 $1 $2 $3 $4 $5 512 $stem
  tools/bin/ring_probe_ps_2 $1 $2 $3 $4 $5 512 $stem
done
