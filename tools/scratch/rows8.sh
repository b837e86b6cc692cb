bash tools/build_variant.sh norm.hip gpurun_out/variants/rows8.so -DRDFC_NORM_ROWS8 || exit 1
for sh in "8 57 76 256" "8 114 152 128" "8 228 304 64"; do
  echo -n "base  $sh: "; bash tools/norm_cold.sh $sh 2>/dev/null | grep bwd_reduce | awk '{print $5, $6}'
  echo -n "rows8 $sh: "; RDFC_HIP_LIB=gpurun_out/variants/rows8.so bash tools/norm_cold.sh $sh 2>/dev/null | grep bwd_reduce | awk '{print $5, $6}'
done
