#!/bin/bash
# kernel-trace of a few timed steps; prints union-busy time (>=1 kernel running), idle time and concurrency
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/prof_gap
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gap -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-events "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_gap/*/*kernel_trace.csv')[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Stream_Id']) for r in csv.DictReader(open(f))), key=lambda r: r[0])
n = len(rows); rows = rows[int(n * 0.5):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for s, e, _, _ in rows: ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; wsum = 0; cur = 0; last = ev[0][0]
for t, d in ev:
    if cur > 0: busy += t - last
    wsum += cur * (t - last)
    cur += d; last = t
span = t1 - t0
print(f'kernels {len(rows)}  span {span/1e6:.2f} ms  union-busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %)  idle {(span-busy)/1e6:.2f} ms  sum of kernel time {wsum/1e6:.2f} ms  mean concurrency while busy {wsum/busy:.2f}')
import collections
per = collections.Counter()
for s, e, k, st in rows: per[st] += e - s
print('kernel time per stream (ms):', {k: round(v/1e6, 1) for k, v in per.most_common()})
PY
rm -rf gpurun_out/prof_gap
