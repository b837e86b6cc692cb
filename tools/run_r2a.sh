set -x
mkdir -p gpurun_out/r2a
timeout 600 python -m pytest tests/test_modules_gpu.py -k "graph_mode or train_step or optim" -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench_eager.json 2> gpurun_out/r2a/bench_eager.err; echo "rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --graph --no-cpu-baseline > gpurun_out/r2a/bench_graph_ms.json 2> gpurun_out/r2a/bench_graph_ms.err; echo "rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --graph --graph-single-stream --no-cpu-baseline > gpurun_out/r2a/bench_graph_ss.json 2> gpurun_out/r2a/bench_graph_ss.err; echo "rc=$?"
tail -3 gpurun_out/r2a/pytest.log
for f in eager graph_ms graph_ss; do python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r2a/bench_$f.json').read().strip().splitlines()[-1])
    print('$f', d['value'], d['ms_per_step'], d.get('unsynced_ms_per_step'), d.get('host_enqueue_ms_per_step'), d.get('launch'), d.get('cpu_baseline'))
except Exception as e:
    print('$f', 'ERR', e)
PY
tail -3 gpurun_out/r2a/bench_$f.err
done
