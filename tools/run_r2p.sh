mkdir -p gpurun_out/r2p
timeout 900 python -m pytest tests/test_variants_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "FUSED_STATS or WGRAD_GLDS or stacked" > gpurun_out/r2p/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2p/pytest.log
for v in 0 1; do
  if [ $v = 1 ]; then export RDFC_FUSED_STATS=1; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_stats=$v', d['value'], d['ms_per_step'], d['unsynced_ms_per_step'])"
done
