mkdir -p gpurun_out/r2m
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv_fwd_dgrad_wgrad" > gpurun_out/r2m/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2m/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --by-shape gpurun_out/r2m/by_shape.txt > gpurun_out/r2m/bench.json 2> gpurun_out/r2m/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2m/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['unsynced_ms_per_step'], 'host', d['host_enqueue_ms_per_step'])
PY
grep "head_dgrad\|igemm_kernel<__bf16,128,128,2,false" gpurun_out/r2m/by_shape.txt
