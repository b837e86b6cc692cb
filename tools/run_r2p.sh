#!/bin/bash
mkdir -p gpurun_out
python tools/pack_misses.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "batched_weight_pack" 2>&1 | tail -3
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
run batchpack $B
RDFC_BATCH_PACK=0 run nobatchpack $B
run batchpack2 $B
RDFC_BATCH_PACK=0 run nobatchpack2 $B
