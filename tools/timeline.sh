#!/bin/bash
# kernel-trace of a few timed steps -> where a step's wall time goes: idle, one kernel alone (by kernel), overlapped
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/prof_tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-events "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob('gpurun_out/prof_tl/*/*kernel_trace.csv')[0]
R = list(csv.DictReader(open(f)))
print('columns:', list(R[0].keys()))
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id')),
                int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1),
                int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1))
               for r in R), key=lambda r: r[0])
# steady state: find adam_kernel launches -> step boundaries
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2] or 'adam_scaled_kernel' in r[2]]
# two adam launches per step (G, D): take the steps between the 4th-last pair and the last pair
ends = adam[1::2]
a, b = ends[-4], ends[-1]
rows = rows[a + 1:b + 1]
nsteps = 3
def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    k = re.sub(r'^void ', '', k)
    return k.split('(')[0][:60]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for i, (s, e, *_rest) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active = set(); last = ev[0][0]
idle = 0; alone = collections.Counter(); over = 0; alone_t = 0
conc = collections.Counter()
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        n = len(active)
        conc[min(n, 4)] += dt
        if n == 0: idle += dt
        elif n == 1:
            k = rows[next(iter(active))]
            alone[short(k[2])] += dt; alone_t += dt
        else: over += dt
    if d > 0: active.add(i)
    else: active.discard(i)
    last = t
span = (t1 - t0) / nsteps
print(f'per step: span {span/1e6:.2f} ms, idle {idle/nsteps/1e6:.2f}, one kernel alone {alone_t/nsteps/1e6:.2f}, two or more {over/nsteps/1e6:.2f}')
print('time by number of running kernels (ms/step):', {k: round(v/nsteps/1e6, 2) for k, v in sorted(conc.items())})
print('alone, by kernel (ms/step):')
for k, v in alone.most_common(30):
    print(f'   {v/nsteps/1e6:7.3f}  {k}')
per = collections.Counter()
for s, e, k, st, *_ in rows: per[st] += e - s
print('kernel time per stream (ms/step):', {k: round(v/nsteps/1e6, 1) for k, v in per.most_common()})
fam = collections.defaultdict(collections.Counter)
for s, e, k, st, *_ in rows: fam[st][short(k)[:44]] += e - s
for st, _v in per.most_common():
    print(f'stream {st}: kernel time by kernel (ms/step)')
    for k, v in fam[st].most_common(14):
        print(f'   {v/nsteps/1e6:7.3f}  {k}')
# per-millisecond picture of the LAST step: busy fraction of every stream and the kernel with the most time in the bin
a2, b2 = ends[-2], ends[-1]
last_rows = [r for r in rows if r[0] >= rows[0][0]][-(b2 - a2):]
s0 = last_rows[0][0]
streams = sorted({r[3] for r in last_rows})
nb = int((max(r[1] for r in last_rows) - s0) / 1e6) + 1
print('last step, 1-ms bins: busy fraction per stream ' + str(streams) + ' | top kernel')
for bi in range(nb):
    lo, hi = s0 + bi * 1e6, s0 + (bi + 1) * 1e6
    fr = collections.Counter(); top = collections.Counter()
    for s_, e_, k, st, *_ in last_rows:
        o = min(e_, hi) - max(s_, lo)
        if o > 0:
            fr[st] += o; top[short(k)[:34]] += o
    print(f'{bi:3d} ' + ' '.join(f'{fr[st]/1e6:4.2f}' for st in streams) + ' | ' + ', '.join(f'{k} {v/1e6:.2f}' for k, v in top.most_common(3)))
# small-grid kernels: total time (they cannot fill the chip whatever runs beside them)
small = collections.Counter()
for s, e, k, st, g, wg in rows:
    nwg = g // max(wg, 1)
    if nwg < 256: small[short(k) + f' [{nwg} wg]'] += e - s
print('kernels with < 256 workgroups (ms/step):')
for k, v in small.most_common(15):
    print(f'   {v/nsteps/1e6:7.3f}  {k}')
PY
rm -rf gpurun_out/prof_tl
