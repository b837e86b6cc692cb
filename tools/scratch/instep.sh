bash tools/build_variant.sh conv_igemm_glds.hip gpurun_out/variants/igr4.so -DRDFC_IGEMM_GLDS_RING=4 || exit 1
run() { python bench.py --steps 20 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
echo -n "ig3 wg768: "; run
echo -n "ig3 wg512: "; RDFC_WGRAD_SLOTS=512 run
echo -n "ig4 wg768: "; RDFC_HIP_LIB=gpurun_out/variants/igr4.so run
echo -n "ig4 wg512: "; RDFC_WGRAD_SLOTS=512 RDFC_HIP_LIB=gpurun_out/variants/igr4.so run
done
