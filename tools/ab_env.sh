#!/bin/bash
# A/B an environment switch on conv shapes (same library, same box, interleaved): tools/ab_env.sh VAR=val
for rep in 1 2; do
for sh in "8 228 304 64 64 3 1 1 fwd" "8 228 304 64 64 3 1 1 dgrad" "8 57 76 256 256 3 1 1 fwd" "8 114 152 128 128 3 1 1 fwd" "8 29 38 512 512 3 1 1 fwd" "8 228 304 128 64 3 1 1 fwd" "8 228 304 128 32 3 1 1 fwd" "8 59 78 256 256 3 1 0 fwd" "8 228 304 128 1 3 1 1 fwd"; do
  a=$(python tools/conv_bench.py $sh 2>/dev/null | awk '{print $(NF-1)}')
  b=$(env $1 python tools/conv_bench.py $sh 2>/dev/null | awk '{print $(NF-1)}')
  echo "$sh : base=$a $1=$b TFLOP/s"
done; done
