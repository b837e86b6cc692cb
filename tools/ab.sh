#!/bin/bash
# A/B two kernel libraries on a list of conv shapes, interleaved, same box.
# usage: tools/ab.sh libA.so libB.so
A=$1; B=$2
for rep in 1 2; do
for sh in "8 228 304 64 128 3 2 1 fwd" "8 228 304 64 128 3 2 1 dgrad" "8 28 38 256 512 4 1 1 fwd" "8 28 38 256 512 4 1 1 dgrad" "8 114 152 192 384 1 1 0 fwd" "8 57 76 128 256 4 2 1 fwd" "8 114 152 128 256 3 2 1 dgrad"; do
  a=$(RDFC_HIP_LIB=$A python tools/conv_bench.py $sh 2>/dev/null | awk '{print $(NF-1)}')
  b=$(RDFC_HIP_LIB=$B python tools/conv_bench.py $sh 2>/dev/null | awk '{print $(NF-1)}')
  echo "$sh : A=$a B=$b TFLOP/s"
done; done
