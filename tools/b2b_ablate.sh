#!/bin/bash
cd $GRAFT_REPO_ROOT
B="hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Wno-inline-asm -I deep-online-video-stabilization_amd/csrc"
for m in 0 1 2; do $B -DB2B_ABLATE=$m -o /tmp/b2b_a$m tools/b2b_probe.hip 2>&1 | grep -v warning | head -5; done
for args in "180 320 64" "90 160 128" "180 320 64 2"; do
  for m in 0 1 2; do timeout -k 5 60 /tmp/b2b_a$m $args; done
done
