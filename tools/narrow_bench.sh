#!/bin/bash
# usage (GPU box, repo root): bash tools/narrow_bench.sh   -> us per launch of the four narrow-map launches of NLSPN's guidance path at the
# benchmark's size (16 x 228 x 304), alone on the chip (tools/conv_bench.py; kernel name from rdfc_last_kernel in the step's by-shape table)
for spec in "8 24 fwd" "8 24 dgrad" "8 24 wgrad" "128 8 dgrad" "128 8 fwd" "128 8 wgrad"; do
  set -- $spec
  python3 tools/conv_bench.py 16 228 304 $1 $2 3 1 1 $3 30
done
