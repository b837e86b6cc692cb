#!/bin/bash
for rep in 1 2; do
for sh in "8 228 304 64 64 3 1 1 wgrad" "8 114 152 128 128 3 1 1 wgrad" "8 57 76 256 256 3 1 1 wgrad" "8 59 78 256 256 3 1 0 wgrad" "8 29 38 512 512 3 1 1 wgrad" "8 228 304 128 64 3 1 1 wgrad" "8 228 304 128 32 3 1 1 wgrad"; do
  a=$(python tools/conv_bench.py $sh 2>/dev/null | awk '{print $(NF-1)}')
  b=$(env $1 python tools/conv_bench.py $sh 2>/dev/null | awk '{print $(NF-1)}')
  echo "$sh : default=$a $1=$b TFLOP/s"
done; done
