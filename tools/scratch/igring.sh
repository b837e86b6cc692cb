bash tools/build_variant.sh conv_igemm_glds.hip gpurun_out/variants/igr3.so -DRDFC_IGEMM_GLDS_RING=3 || exit 1
RDFC_HIP_LIB=gpurun_out/variants/igr3.so timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "(test_conv_fwd_dgrad_wgrad or conv_transpose) and bf16" 2>&1 | tail -1
for sh in "8 114 152 64 128 3 2 1" "8 57 76 128 256 3 2 1" "8 228 304 64 128 4 2 1" "8 114 152 128 256 4 2 1" "8 29 38 256 512 3 2 1"; do
  for w in fwd dgrad; do
  echo -n "ring4: "; python tools/conv_bench.py $sh $w 30
  echo -n "ring3: "; RDFC_HIP_LIB=gpurun_out/variants/igr3.so python tools/conv_bench.py $sh $w 30
  done
done
