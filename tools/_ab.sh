python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -3
python -m pytest tests/test_modules_gpu.py -x -q 2>&1 | tail -3
b() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-as-worded --no-kernel-events 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"; }
b fused; RDFC_FUSED_STATS=0 b nofuse; b fused; RDFC_FUSED_STATS=0 b nofuse
