#!/bin/bash
# kernel-trace of a few graph-replayed steps; prints per-step busy time vs idle gaps between consecutive kernels
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/prof_gap
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gap -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-events "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_gap/*/*kernel_trace.csv')[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# last 40% of the trace = steady-state timed steps
n = len(rows); rows = rows[int(n * 0.5):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
pos = [g for g in gaps if g > 0]
print(f'kernels {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms  mean gap {sum(pos)/max(1,len(pos))/1e3:.2f} us  overlapped pairs {sum(1 for g in gaps if g < 0)}')
import collections
short = collections.Counter()
for s, e, k in rows:
    if e - s < 8000: short[k[:60]] += 1
print('kernels shorter than 8 us:', sum(short.values()), short.most_common(8))
hist = collections.Counter(min(int(g / 1000), 20) for g in pos)
print('gap histogram (us: count):', sorted(hist.items()))
PY
rm -rf gpurun_out/prof_gap
