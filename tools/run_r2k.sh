# A/B of the conv3x3_glds tile variants at the batched step's sizes (B = 16), one stream, per-shape table
mkdir -p gpurun_out/r2k
for mode in 1 3 6; do
RDFC_GLDS=$mode RDFC_BRANCH_STREAMS=0 RDFC_NET_STREAMS=0 RDFC_WGRAD_STREAM=0 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --event-every 1 --by-shape gpurun_out/r2k/by_shape_mode$mode.txt > gpurun_out/r2k/bench_mode$mode.json 2> gpurun_out/r2k/err$mode.log
echo "== RDFC_GLDS=$mode (one stream)"; python - gpurun_out/r2k/bench_mode$mode.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['unsynced_ms_per_step'])
PY
grep "conv3x3_glds" gpurun_out/r2k/by_shape_mode$mode.txt | head -12
RDFC_GLDS=$mode timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > gpurun_out/r2k/bench3s_mode$mode.json 2>/dev/null
python - gpurun_out/r2k/bench3s_mode$mode.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('3 streams:', d['value'], d['ms_per_step'], d['unsynced_ms_per_step'])
PY
done
