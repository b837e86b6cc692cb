bash tools/build_variant.sh conv_wgrad_glds.hip gpurun_out/variants/wgr3.so -DRDFC_WGRAD_GLDS_RING=3 || exit 1
RDFC_HIP_LIB=gpurun_out/variants/wgr3.so timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "(test_conv_fwd_dgrad_wgrad or conv_transpose) and bf16" 2>&1 | tail -1
for sh in "8 57 76 256 256 3 1 1" "8 114 152 128 128 3 1 1" "8 228 304 64 64 3 1 1" "8 29 38 512 512 3 1 1" "8 114 152 256 128 3 1 1" "8 228 304 128 64 3 1 1"; do
  echo -n "ring4 512: "; python tools/conv_bench.py $sh wgrad 30
  echo -n "ring4 768: "; RDFC_WGRAD_SLOTS=768 python tools/conv_bench.py $sh wgrad 30
  echo -n "ring3 512: "; RDFC_HIP_LIB=gpurun_out/variants/wgr3.so python tools/conv_bench.py $sh wgrad 30
  echo -n "ring3 768: "; RDFC_WGRAD_SLOTS=768 RDFC_HIP_LIB=gpurun_out/variants/wgr3.so python tools/conv_bench.py $sh wgrad 30
done
