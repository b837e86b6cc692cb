# quick GPU check: the step-level tests + the default bench (no CPU baseline)
tag=${1:-quick}
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "train_step or graph or combine or fanout or elementwise or stacked or dp or g_a2b or g_b2a or patchgan" > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/$tag/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${@:2} > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; echo "bench rc=$?"
python - "gpurun_out/$tag/bench.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['unsynced_ms_per_step'], d['host_enqueue_ms_per_step'], d['launch'])
    print(d.get('roofline'))
except Exception as e: print('ERR', e)
PY
grep -v amdgpu.ids gpurun_out/$tag/bench.err | tail -5
