for shape in "1 180 320 64 64 3 1 1" "1 90 160 128 128 3 1 1" "1 45 80 256 256 3 1 1" "1 23 40 512 512 3 1 1" "1 45 80 1024 256 1 1 0" "1 180 320 64 256 1 1 0" "1 180 320 256 64 1 1 0"; do
 for t in 0 1 2; do for sk in 1 2 4 8; do
  echo -n "shape=[$shape] tile=$t splitk=$sk : "; STABNET_CONV_TILE=$t STABNET_CONV_SPLITK=$sk python tools/conv_bench.py $shape 30 2>&1 | tail -1
 done; done
done
