for i in 1 2 3; do
for f in 0 1; do
RDFC_DA_EARLY=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --no-as-worded 2>gpurun_out/err_$f.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('da_early=$f', d['value'], d['ms_per_step'])"
done
done
tail -5 gpurun_out/err_1.log
