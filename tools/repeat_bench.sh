#!/bin/bash
# usage: tools/repeat_bench.sh [n] [bench flags]   -> value / ms per step / unsynced / host enqueue / in-step frac of n bench runs in a row
n=${1:-6}; shift
mkdir -p gpurun_out
for i in $(seq 1 $n); do
python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['unsynced_ms_per_step'], d['host_enqueue_ms_per_step'], d['roofline']['frac'], d['roofline']['isolated']['frac'])"
done
