#!/bin/bash
# per-phase timeline of single conv launches at the in-network shapes (tools/ring_probe.hip with RING_STAMP)
cd $GRAFT_REPO_ROOT
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -DRING_STAMP=1 -Wno-inline-asm -I deep-online-video-stabilization_amd/csrc -o /tmp/ring_stamp tools/ring_probe.hip 2>&1 | grep -v warning | head -5
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Wno-inline-asm -I deep-online-video-stabilization_amd/csrc -o /tmp/ring_plain tools/ring_probe.hip 2>&1 | grep -v warning | head -5
for args in "60 60 256 1024 1" "60 60 2048 256 1" "60 60 512 256 1" "60 60 256 256 3" "120 120 128 512 1" "240 240 64 256 1" "240 240 64 64 3" "23 40 512 2048 1" "23 40 512 512 3"; do
  /tmp/ring_plain $args
  /tmp/ring_stamp $args
done
