#!/bin/bash
# usage: tools/ab_libs.sh "shape1;shape2;..." lib1.so lib2.so ...   (interleaved, two repetitions, TFLOP/s)
IFS=';' read -ra SH <<< "$1"; shift
for rep in 1 2; do
for sh in "${SH[@]}"; do
  line="$sh :"
  for lib in "$@"; do
    v=$(RDFC_HIP_LIB=$lib python tools/conv_bench.py $sh 30 2>/dev/null | awk '{print $(NF-1)}')
    line="$line $(basename $lib .so)=$v"
  done
  echo "$line"
done; done
