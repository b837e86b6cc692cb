python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py tests/test_fold_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --no-as-worded 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('now', d['value'], d['ms_per_step'])"
done
