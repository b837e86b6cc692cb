mkdir -p gpurun_out/r2h
timeout 1500 python -m pytest tests -m gpu -q -s -x > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2h/pytest.log
grep -n "full-size step" gpurun_out/r2h/pytest.log | cut -c1-900
for cfg in "bf16:--dtype bf16" "f16_480:--dtype f16 --height 480 --width 640 --batch 4" "f16:--dtype f16" "f32:--dtype f32 --steps 5 --warmup 2"; do
tag=${cfg%%:*}; flags=${cfg#*:}
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events $flags > gpurun_out/r2h/bench_$tag.json 2> gpurun_out/r2h/bench_$tag.err; echo "bench $tag rc=$?"
python - "gpurun_out/r2h/bench_$tag.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['unsynced_ms_per_step'], d['step_tflops'], d['step_frac_of_mfma_peak'], d['config']['workload'][-60:])
except Exception as e: print('ERR', e)
PY
grep -v amdgpu.ids gpurun_out/r2h/bench_$tag.err | tail -3
done
