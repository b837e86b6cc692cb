RDFC_GLDS=4 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "(test_conv_fwd_dgrad_wgrad or conv_transpose) and bf16" 2>&1 | tail -1
for sh in "2 57 76 256 256 3 1 1" "8 57 76 256 256 3 1 1" "16 57 76 256 256 3 1 1" "8 114 152 128 128 3 1 1" "8 228 304 64 64 3 1 1" "8 29 38 512 512 3 1 1" "8 114 152 256 128 3 1 1"; do
  for w in fwd; do
    echo -n "base: "; python tools/conv_bench.py $sh $w 30
    echo -n "spec: "; RDFC_GLDS=4 python tools/conv_bench.py $sh $w 30
  done
done
RDFC_GLDS=4 bash tools/glds_stamps.sh 2 57 76 256 256 2>&1 | grep "per tap" | head -2
