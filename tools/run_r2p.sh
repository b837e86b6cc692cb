#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_modules_gpu.py tests/test_dp_gpu.py -x -q -m gpu -k "trace or full_size or rccl or ranks" 2>&1 | tail -3
run() { tag=$1; shift; "$@" > gpurun_out/r2p_$tag.json 2> gpurun_out/r2p_$tag.err; python - $tag <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(f'gpurun_out/r2p_{sys.argv[1]}.json') if l.startswith('{"metric"')][-1])
    print(sys.argv[1], d['value'], d['ms_per_step'], d['unsynced_ms_per_step'], d['host_enqueue_ms_per_step'], d['roofline']['frac'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open(f'gpurun_out/r2p_{sys.argv[1]}.err').read()[-1500:])
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
for rep in 1 2 3; do
run early_da $B
RDFC_EARLY_D_A=0 run late_da $B
done
run rccl $B --rccl-self-test
tail -1 gpurun_out/r2p_rccl.json | cut -c1-60
