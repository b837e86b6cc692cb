#!/bin/bash
# kernel-time table of the widened step (bench.py --label-net) minus the table of the headline step: what G_label adds
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
for tag in base label; do
  rm -rf gpurun_out/prof_$tag
  flag=""; [ $tag = label ] && flag="--label-net"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events $flag > gpurun_out/prof_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob
def load(tag):
    f = glob.glob(f'gpurun_out/prof_{tag}/*/*kernel_stats.csv')[0]
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(f))}
b, l = load('base'), load('label')
steps = 6 + 2        # --steps 6 --warmup 2, --no-kernel-events: no extra isolated step
rows = []
for k, (n, t) in l.items():
    n0, t0 = b.get(k, (0, 0.0))
    rows.append(((t - t0) / steps / 1e6, (n - n0) / steps, k))
tot = sum(r[0] for r in rows)
print(f'G_label adds {tot:.2f} ms of kernel time per step')
for ms, n, k in sorted(rows, reverse=True)[:22]:
    print(f'{ms:7.3f} ms  {n:6.1f} launches  {1e3*ms/max(n,1e-9):8.1f} us  {k[:100]}')
PY
rm -rf gpurun_out/prof_base gpurun_out/prof_label
