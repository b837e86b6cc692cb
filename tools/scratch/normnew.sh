python -m pytest tests/test_kernels_gpu.py -q -x -k "norm or adain or stat or chsum" 2>&1 | tail -2
for sh in "8 57 76 256" "8 114 152 128" "8 228 304 64" "8 57 76 256 8"; do echo "== $sh"; bash tools/norm_cold.sh $sh 2>/dev/null | grep -E "kernel|bwd_" | awk '{print "  ", $1, $5, $6}'; done
