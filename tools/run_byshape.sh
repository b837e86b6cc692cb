mkdir -p gpurun_out/r2e
RDFC_BRANCH_STREAMS=0 RDFC_NET_STREAMS=0 RDFC_WGRAD_STREAM=0 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --event-every 1 --by-shape gpurun_out/r2e/by_shape_1stream.txt > gpurun_out/r2e/bench_1stream.json 2> gpurun_out/r2e/err.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2e/bench_1stream.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['unsynced_ms_per_step'])
PY
head -70 gpurun_out/r2e/by_shape_1stream.txt
