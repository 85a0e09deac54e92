#!/bin/bash
# Where do the low-K 1x1 launches of block1/2 (K = 64 / 128) spend their time?  plain / no MFMA / no DMA / stamps, several grids.
cd $GRAFT_REPO_ROOT
for m in 0 8 1; do hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -DRING_ABLATE=$m -Wno-inline-asm -I deep-online-video-stabilization_amd/csrc -o /tmp/ring_a$m tools/ring_probe.hip 2>&1 | grep -v warning | grep error; done
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -DRING_STAMP=1 -Wno-inline-asm -I deep-online-video-stabilization_amd/csrc -o /tmp/ring_stamp tools/ring_probe.hip 2>&1 | grep error
for args in "180 320 64 320 1" "180 320 64 256 1" "90 160 128 512 1" "90 160 64 256 1"; do
  for wg in 768 512 1024 1280; do /tmp/ring_a0 $args $wg; done
  /tmp/ring_a8 $args 768; /tmp/ring_a1 $args 768; /tmp/ring_stamp $args 768
done
