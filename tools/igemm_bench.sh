#!/bin/bash
# igemm_glds (default) vs the register-staged igemm
for args in "8 28 38 256 512 4 1 1 fwd" "8 28 38 256 512 4 1 1 dgrad" "8 228 304 64 128 3 2 1 dgrad" "8 228 304 64 128 3 2 1 fwd" "8 114 152 128 256 3 2 1 dgrad" "8 114 152 128 256 3 2 1 fwd" "8 114 152 192 384 1 1 0 fwd" "8 57 76 128 256 4 2 1 dgrad" "8 57 76 128 256 4 2 1 fwd" "8 228 304 64 192 3 2 1 dgrad" "8 29 38 512 512 3 2 1 dgrad" "8 29 38 512 512 3 2 1 fwd" "8 57 76 256 512 1 2 0 dgrad" "8 57 76 256 512 3 2 1 fwd"; do
  python tools/conv_bench.py $args 30
  RDFC_IGEMM_GLDS=0 python tools/conv_bench.py $args 30
done
