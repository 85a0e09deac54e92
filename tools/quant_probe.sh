for M in 8192 12288 14400 16384 20480 24576 28672 32768 49152; do python tools/conv_bench.py 1 1 $M 1152 128 1 1 0 100 0; done
echo K=256
for M in 8192 16384 24576 32768 57600 65536; do python tools/conv_bench.py 1 1 $M 256 128 1 1 0 100 0; done
