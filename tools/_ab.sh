for v in NOREAD NOMFMA NODMA; do bash tools/build_variant.sh conv_igemm_glds.hip gpurun_out/variants/i$v.so -DRDFC_EXP_$v || exit 1; done
for shp in "16 114 152 128 256 3 2 1" "16 28 38 256 512 4 1 1" "16 114 152 192 384 1 1 0" "16 228 304 64 128 3 2 1"; do
  python tools/conv_bench.py $shp fwd 30 2>/dev/null | tail -1
  for v in NOREAD NOMFMA NODMA; do RDFC_HIP_LIB=gpurun_out/variants/i$v.so python tools/conv_bench.py $shp fwd 30 2>/dev/null | tail -1 | sed "s/^/   $v: /"; done
done
