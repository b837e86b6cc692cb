#!/bin/bash
for args in "8 228 304 64 128 3 2 1 wgrad" "8 114 152 128 256 3 2 1 wgrad" "8 57 76 256 512 3 2 1 wgrad" "8 28 38 256 512 4 1 1 wgrad" "8 114 152 192 384 1 1 0 wgrad" "8 57 76 128 256 4 2 1 wgrad" "8 228 304 64 192 3 2 1 wgrad" "8 114 152 64 128 4 2 1 wgrad" "8 57 76 384 768 1 1 0 wgrad"; do
  python tools/conv_bench.py $args 30
  RDFC_WGRAD_GATHER_GLDS=0 python tools/conv_bench.py $args 30
done
