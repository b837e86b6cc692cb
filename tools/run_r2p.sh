#!/bin/bash
mkdir -p gpurun_out
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
for rep in 1 2 3 4; do
run plain $B
run rccl $B --rccl-self-test
done
run nlspn $B --nlspn
run f16big $B --dtype f16 --height 480 --width 640 --batch 4
