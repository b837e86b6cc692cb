#!/bin/bash
for args in "8 234 310 64 3 7 1 0 dgrad" "8 228 304 128 1 3 1 1 dgrad" "8 228 304 96 1 3 1 1 dgrad" "8 234 310 1 64 7 1 0 fwd" "8 228 304 3 64 3 1 1 fwd" "8 228 304 3 64 4 2 1 fwd" "8 228 304 3 48 3 1 1 fwd" "8 228 304 1 64 4 2 1 fwd" "8 228 304 1 16 3 1 1 fwd"; do
  python tools/conv_bench.py $args 30
  RDFC_NO_THIN_IN=1 python tools/conv_bench.py $args 30
done
