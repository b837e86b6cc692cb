for i in 1 2 3; do
for f in "1 4" "2 4" "2 8" "1 8"; do
set -- $f
RDFC_WGRAD_STREAMS=$1 GPU_MAX_HW_QUEUES=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --no-as-worded 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wstreams=$1 hwq=$2', d['value'], d['ms_per_step'])"
done
done
