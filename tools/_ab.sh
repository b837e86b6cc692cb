python -m pytest tests/test_kernels_gpu.py -x -q -k "norm or adain or bn_over or fused_stat or channel or chsum or small_elem" 2>&1 | tail -2
for i in 1 2 3; do
for f in 32 0 8 4; do
RDFC_NORM_FC=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --no-as-worded 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fc=$f', d['value'], d['ms_per_step'])"
done
done
