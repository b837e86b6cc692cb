#!/bin/bash
mkdir -p gpurun_out
RDFC_D_UNDER_G=first python -m pytest tests/test_modules_gpu.py -x -q -m gpu -k "trace or full_size" 2>&1 | tail -3
run() { tag=$1; shift; "$@" > gpurun_out/r2p_$tag.json 2> gpurun_out/r2p_$tag.err; python - $tag <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(f'gpurun_out/r2p_{sys.argv[1]}.json') if l.startswith('{"metric"')][0])
    print(sys.argv[1], d['value'], d['ms_per_step'], d['unsynced_ms_per_step'], d['host_enqueue_ms_per_step'], d['roofline']['frac'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open(f'gpurun_out/r2p_{sys.argv[1]}.err').read()[-1500:])
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
for rep in 1 2 3; do
RDFC_D_UNDER_G=first run d_first $B
RDFC_D_UNDER_G=1 run d_under_g $B
RDFC_D_UNDER_G=0 run d_after_g $B
done
