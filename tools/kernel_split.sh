#!/bin/bash
# kernel-time split of one train step, single stream (no overlap: durations are what each kernel needs alone, inputs as cold as in the step)
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/prof_ks
RDFC_BRANCH_STREAMS=0 RDFC_NET_STREAMS=0 RDFC_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ks -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-events "$@" > gpurun_out/kernel_split.log 2>&1
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('gpurun_out/prof_ks/*/*kernel_stats.csv')[0])))
steps = 8 + 2 + 0
def tot(pred): return sum(float(r['TotalDurationNs']) for r in rows if pred(r['Name'])) / 1e6 / steps
fam = dict(norm=lambda n: any(k in n for k in ('stats_kernel', 'apply_kernel', 'bwd_reduce', 'bwd_apply', 'finalize', 'adain', 'chsum')),
           wgrad=lambda n: 'wgrad' in n,
           conv=lambda n: any(k in n for k in ('conv3x3', 'igemm', 'thin', 'slab', 'patch')) and 'wgrad' not in n,
           torch=lambda n: 'at::native' in n)
allk = tot(lambda n: True)
parts = {k: tot(f) for k, f in fam.items()}
print(f'kernel time per step, single stream: {allk:.1f} ms  ' + '  '.join(f'{k} {v:.1f}' for k, v in parts.items()) + f'  rest {allk - sum(parts.values()):.1f}')
for r in rows[:12]:
    print('  %5.1f%% %5d/step %7.1f us  %s' % (float(r['Percentage']), int(r['Calls']) // steps, float(r['AverageNs']) / 1e3, r['Name'][:80]))
PY
rm -rf gpurun_out/prof_ks
