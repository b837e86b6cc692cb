mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2c/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r2c/pytest.log
for mode in "" "--graph"; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $mode > gpurun_out/r2c/bench$mode.json 2> gpurun_out/r2c/bench$mode.err; echo "bench $mode rc=$?"
python - "gpurun_out/r2c/bench$mode.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['unsynced_ms_per_step'], d['host_enqueue_ms_per_step'], d['launch'])
    for k,v in d.get('kernels',{}).items(): print('   ', k, v)
except Exception as e: print('ERR', e)
PY
grep -v amdgpu.ids gpurun_out/r2c/bench$mode.err | tail -5
done
