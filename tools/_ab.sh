for i in 1 2; do
for f in 0 256 384 512 1024; do
RDFC_WGRAD_SLOTS=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --no-as-worded 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slots=$f', d['value'], d['ms_per_step'])"
done
done
