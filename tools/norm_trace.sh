#!/bin/bash
# per-dispatch durations of the normalisation kernels inside the train step, grouped by grid size
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/prof_nt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_nt -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_nt/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
cols = rows[0].keys()
gk = [c for c in cols if 'Grid_Size' in c or c == 'Grid_Size_X']
print('columns:', [c for c in cols][:30])
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[len(rows)//2:]:
    n = r['Kernel_Name']
    for key in ('bwd_apply_kernel', 'bwd_reduce_kernel', 'apply_kernel', 'stats_kernel'):
        if key in n and ('bwd_' + key not in n):
            g = (r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'))
            a = agg[(key, g)]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            break
seen = {}
for r in rows:
    n = r['Kernel_Name']
    if ('_kernel' in n) and n not in seen:
        seen[n] = (r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'], r['Scratch_Size'], r['LDS_Block_Size'])
for n, v in sorted(seen.items()):
    if any(k in n for k in ('apply', 'reduce', 'stats', 'chsum')):
        print('vgpr/agpr/sgpr/scratch/lds', v, n[:100])
for (key, g), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f'{key:18s} grid {g}  calls {n:4d}  total {t/1e3:7.2f} ms  avg {t/n:7.1f} us')
PY
rm -rf gpurun_out/prof_nt
