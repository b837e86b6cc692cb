bash tools/build_variant.sh norm.hip gpurun_out/variants/noconst.so -DRDFC_NORM_EXP_NOCONST || exit 1
bash tools/build_variant.sh norm.hip gpurun_out/variants/notail.so -DRDFC_NORM_EXP_NOTAIL || exit 1
bash tools/build_variant.sh norm.hip gpurun_out/variants/neither.so -DRDFC_NORM_EXP_NOTAIL -DRDFC_NORM_EXP_NOCONST || exit 1
for sh in "8 57 76 256" "8 114 152 128"; do
  echo -n "base    $sh: "; bash tools/norm_cold.sh $sh 2>/dev/null | grep bwd_reduce | awk '{print $5, $6}'
  for v in noconst notail neither; do
  echo -n "$v $sh: "; RDFC_HIP_LIB=gpurun_out/variants/$v.so bash tools/norm_cold.sh $sh 2>/dev/null | grep bwd_reduce | awk '{print $5, $6}'
  done
done
