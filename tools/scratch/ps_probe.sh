#!/bin/bash
for geo in "90 160 128 128 3" "45 80 512 1280 1" "720 1280 13 64 7"; do
  set -- $geo
  stem=0; [ "$5" = "7" ] && stem=1
  echo "== H=$1 W=$2 Cin=$3 N=$4 KH=$5"
  tools/bin/ring_probe_pk_0 $1 $2 $3 $4 $5 512 $stem
  tools/bin/ring_probe_ps_0 $1 $2 $3 $4 $5 512 $stem
  tools/bin/ring_probe_ps_2 $1 $2 $3 $4 $5 512 $stem
done
