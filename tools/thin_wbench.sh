#!/bin/bash
# A/B of the thin weight-gradient kernel against the per-tap MFMA kernel
for args in "8 234 310 1 64 7 1 0 wgrad" "8 234 310 64 3 7 1 0 wgrad" "8 228 304 128 1 3 1 1 wgrad" "8 228 304 96 1 3 1 1 wgrad" "8 228 304 1 16 3 1 1 wgrad" "8 228 304 3 64 3 1 1 wgrad" "8 228 304 3 48 3 1 1 wgrad"; do
  python tools/conv_bench.py $args 30
  RDFC_NO_WGRAD_THIN=1 python tools/conv_bench.py $args 30
done
