#!/bin/bash
# A/B two kernel libraries on a list of conv shapes, interleaved, same box.
# usage: tools/ab.sh libA.so libB.so
A=$1; B=$2
for rep in 1 2; do
for sh in "8 57 76 256 256 3 1 1 wgrad" "8 114 152 128 128 3 1 1 wgrad" "8 29 38 512 512 3 1 1 wgrad" "8 59 78 256 256 3 1 0 wgrad" "8 228 304 64 128 3 2 1 wgrad"; do
  a=$(RDFC_HIP_LIB=$A python tools/conv_bench.py $sh 2>/dev/null | awk '{print $(NF-1)}')
  b=$(RDFC_HIP_LIB=$B python tools/conv_bench.py $sh 2>/dev/null | awk '{print $(NF-1)}')
  echo "$sh : A=$a B=$b TFLOP/s"
done; done
