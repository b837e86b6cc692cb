python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
for i in 1 2; do
for f in "0 0" "1 0" "0 -1" "1 -1"; do
set -- $f
RDFC_PRIO_WGRAD=$1 RDFC_PRIO_HELPER=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --no-as-worded 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio wgrad=$1 helper=$2', d['value'], d['ms_per_step'])"
done
done
