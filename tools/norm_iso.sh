#!/bin/bash
# per-kernel durations of the normalisation kernels on one map, alone on the GPU: bash tools/norm_iso.sh N H W C G
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/prof_ni
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ni -- python3 tools/norm_bench.py "$@" > gpurun_out/norm_iso.log 2>&1
grep -v "^W\|^E\|^I" gpurun_out/norm_iso.log | tail -8
python3 - <<'PY'
import csv, glob
for r in csv.DictReader(open(glob.glob('gpurun_out/prof_ni/*/*kernel_stats.csv')[0])):
    if any(k in r['Name'] for k in ('apply', 'reduce', 'stats', 'chsum', 'finalize')):
        print(f"{int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  {r['Name'][:80]}")
PY
rm -rf gpurun_out/prof_ni
