bash tools/build_variant.sh norm.hip gpurun_out/variants/skipred.so -DRDFC_EXP_SKIP_BWD_REDUCE || exit 1
b() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-as-worded --no-kernel-events 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"; }
b base; RDFC_HIP_LIB=gpurun_out/variants/skipred.so b skip_bwd_reduce; b base; RDFC_HIP_LIB=gpurun_out/variants/skipred.so b skip_bwd_reduce
