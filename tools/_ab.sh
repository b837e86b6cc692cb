python -m pytest tests/test_fold_gpu.py tests/test_kernels_gpu.py tests/test_variants_gpu.py -x -q 2>&1 | tail -3
b() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"; }
b par; b par; b par
