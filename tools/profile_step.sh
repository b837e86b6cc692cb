#!/bin/bash
# rocprofv3 kernel trace of the bench command (default workload) -> gpurun_out/<tag>/kernel_stats.csv + a family table
tag=${1:-prof}; shift
root=$PWD
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/$tag/raw
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/raw -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events "$@" > gpurun_out/$tag/bench_under_rocprof.json 2> gpurun_out/$tag/err.log
f=$(ls gpurun_out/$tag/raw/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/$tag/kernel_stats.csv
python3 - gpurun_out/$tag/kernel_stats.csv <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
steps = 6 + 2 + 6      # timed + warm-up + the unsynced repeat of the timed steps
print(f'total kernel time {tot/1e6:.1f} ms over {steps} steps = {tot/1e6/steps:.2f} ms per step')
fam = {}
for r in rows:
    n = r['Name']
    k = re.sub(r'<.*', '', n.replace('void ', '').replace('(anonymous namespace)::', ''))
    d = fam.setdefault(k, [0, 0.0]); d[0] += int(r['Calls']); d[1] += float(r['TotalDurationNs'])
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f'{100*t/tot:6.2f} %  {t/1e6/steps:7.3f} ms/step  {c/steps:7.1f} launches/step  {t/c/1e3:8.1f} us  {k[:90]}')
PY
rm -rf gpurun_out/$tag/raw
