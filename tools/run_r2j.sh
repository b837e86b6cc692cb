mkdir -p gpurun_out/r2j
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -m gpu -x -q -k "nlspn or g_b2a or patchgan or full_size" > gpurun_out/r2j/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2j/pytest.log
bash tools/profile_step.sh r2j_nlspn --nlspn 2>&1 | grep -i "total\|prop_\|affinity\|igemm_kernelIDF16bLi16\|slab_kernelIDF16bLi16"
for flag in "--nlspn" ""; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events $flag > gpurun_out/r2j/bench$flag.json 2> gpurun_out/r2j/bench$flag.err; echo "bench $flag rc=$?"
python - "gpurun_out/r2j/bench$flag.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['unsynced_ms_per_step'])
except Exception as e: print('ERR', e)
PY
done
