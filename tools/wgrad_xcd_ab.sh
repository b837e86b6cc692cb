#!/bin/bash
for args in "8 57 76 256 256 3 1 1 wgrad" "8 228 304 64 64 3 1 1 wgrad" "8 114 152 128 128 3 1 1 wgrad" "8 29 38 512 512 3 1 1 wgrad" "8 228 304 128 64 3 1 1 wgrad" "8 114 152 256 128 3 1 1 wgrad"; do
  for rep in 1 2; do
  RDFC_WGRAD_XCD=1 python tools/conv_bench.py $args 30
  RDFC_WGRAD_XCD=0 python tools/conv_bench.py $args 30
  done
done
