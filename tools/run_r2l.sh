mkdir -p gpurun_out/r2l
( time timeout 1800 python -m pytest tests -m gpu -q -x ) > gpurun_out/r2l/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r2l/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2l/bench.json 2> gpurun_out/r2l/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2l/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['unsynced_ms_per_step'], 'host', d['host_enqueue_ms_per_step'])
print(d['roofline'])
PY
