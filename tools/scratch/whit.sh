bash tools/build_variant.sh conv_glds.hip gpurun_out/variants/whit.so -DRDFC_GLDS_EXP_WHIT || exit 1
for sh in "2 57 76 256 256 3 1 1" "8 57 76 256 256 3 1 1" "16 57 76 256 256 3 1 1"; do
  echo -n "base      : "; python tools/conv_bench.py $sh fwd 30
  echo -n "w from L1 : "; RDFC_HIP_LIB=gpurun_out/variants/whit.so python tools/conv_bench.py $sh fwd 30
  echo -n "spec      : "; RDFC_GLDS=4 python tools/conv_bench.py $sh fwd 30
  echo -n "spec w L1 : "; RDFC_GLDS=4 RDFC_HIP_LIB=gpurun_out/variants/whit.so python tools/conv_bench.py $sh fwd 30
done
