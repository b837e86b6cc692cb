#!/bin/bash
# usage (GPU box, repo root): bash tools/ab_env.sh <rounds> "<ENV_A>" "<ENV_B>" [bench flags]
# Interleaved A/B of the train step under two environments (e.g. "" against "RDFC_STREAM_PRIO=wgrad:1"): <rounds> pairs of
# `bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-as-worded` runs on ONE box, pairs/s and ms per step of each printed in order.
rounds=${1:-2}; A="$2"; B="$3"; shift 3
for r in $(seq 1 $rounds); do
  for leg in A B; do
    if [ $leg = A ]; then e="$A"; else e="$B"; fi
    out=$(env $e python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-as-worded "$@" 2>/dev/null | tail -1)
    echo "$leg [$e] $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(d['value'], 'pairs/s', d['ms_per_step'], 'ms; host', d['host_enqueue_ms_per_step'], 'unsynced', d['unsynced_ms_per_step'], 'dominant', d.get('roofline',{}).get('frac'))" "$out")"
  done
done
