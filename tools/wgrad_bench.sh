#!/bin/bash
for args in "8 59 78 256 256 3 1 0 wgrad" "8 57 76 256 256 3 1 1 wgrad" "8 228 304 64 64 3 1 1 wgrad" "8 114 152 128 128 3 1 1 wgrad" "8 29 38 512 512 3 1 1 wgrad" "8 228 304 128 64 3 1 1 wgrad"; do
  python tools/conv_bench.py $args 30
  RDFC_WGRAD_GLDS=0 python tools/conv_bench.py $args 30
done
