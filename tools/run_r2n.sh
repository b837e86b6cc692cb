mkdir -p gpurun_out/r2n
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv_fwd_dgrad_wgrad" > gpurun_out/r2n/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2n/pytest.log
for args in "16 59 78 256 256 3 1 0 wgrad" "16 57 76 256 256 3 1 1 wgrad" "16 114 152 128 128 3 1 1 wgrad" "16 29 38 512 512 3 1 1 wgrad"; do
  python tools/conv_bench.py $args 30
  RDFC_WGRAD_GLDS=4 python tools/conv_bench.py $args 30
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > gpurun_out/r2n/bench.json 2>/dev/null
RDFC_WGRAD_GLDS=4 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > gpurun_out/r2n/bench_old.json 2>/dev/null
python - <<'PY'
import json
for f in ('bench','bench_old'):
    d=json.loads(open(f'gpurun_out/r2n/{f}.json').read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['unsynced_ms_per_step'])
PY
