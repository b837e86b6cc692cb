python -m pytest tests/test_kernels_gpu.py -x -q -k "pointwise_stride2 or fanout" 2>&1 | tail -3
python -m pytest tests/test_modules_gpu.py -x -q 2>&1 | tail -3
for i in 1 2 3; do
for f in 0 1; do
RDFC_POINT_S2=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --no-as-worded 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('point_s2=$f', d['value'], d['ms_per_step'])"
done
done
