mkdir -p gpurun_out/r2g
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "nlspn or mdconv" > gpurun_out/r2g/pytest_nlspn.log 2>&1; echo "nlspn kernel tests rc=$?"; tail -4 gpurun_out/r2g/pytest_nlspn.log
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_label_net_gpu.py tests/test_normal_net_gpu.py -m gpu -q -s > gpurun_out/r2g/pytest_modules.log 2>&1; echo "module tests rc=$?"; tail -4 gpurun_out/r2g/pytest_modules.log
for flag in "--nlspn" ""; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events $flag > gpurun_out/r2g/bench$flag.json 2> gpurun_out/r2g/bench$flag.err; echo "bench $flag rc=$?"
python - "gpurun_out/r2g/bench$flag.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['unsynced_ms_per_step'])
except Exception as e: print('ERR', e)
PY
grep -v amdgpu.ids gpurun_out/r2g/bench$flag.err | tail -5
done
RDFC_NLSPN_STEPWISE=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --nlspn > gpurun_out/r2g/bench_nlspn_stepwise.json 2>/dev/null
python - gpurun_out/r2g/bench_nlspn_stepwise.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('stepwise', d['value'], d['ms_per_step'], d['unsynced_ms_per_step'])
except Exception as e: print('ERR', e)
PY
