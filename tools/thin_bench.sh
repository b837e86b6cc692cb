#!/bin/bash
# A/B of the thin-output conv kernel against the MFMA kernels on the shapes it takes over
for args in "8 27 37 512 1 4 1 1 fwd" "8 234 310 64 3 7 1 0 fwd" "8 234 310 1 64 7 1 0 dgrad" "8 228 304 128 1 3 1 1 fwd" "8 228 304 96 1 3 1 1 fwd" "8 228 304 3 64 3 1 1 dgrad" "8 228 304 1 16 3 1 1 dgrad"; do
  python tools/conv_bench.py $args 30
  RDFC_NO_THIN=1 python tools/conv_bench.py $args 30; RDFC_THIN_TILE_MINPX=100000000 python tools/conv_bench.py $args 30
done
