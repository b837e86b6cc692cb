#!/bin/bash
# per-kernel times of the norm kernels on cache-cold maps: bash tools/norm_cold.sh N H W C [G]
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/prof_nc
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nc -- python3 tools/norm_cold.py "$@" > gpurun_out/norm_cold.log 2>&1
tail -1 gpurun_out/norm_cold.log
python3 - "$@" <<'PY'
import csv, glob, sys
N, H, W, C = (int(v) for v in sys.argv[1:5])
mb = N * H * W * C * 2 / 1e6
traffic = dict(stats_kernel=1, apply_kernel=2, bwd_reduce=2, bwd_apply=3)
for r in csv.DictReader(open(glob.glob('gpurun_out/prof_nc/*/*kernel_stats.csv')[0])):
    for k, t in traffic.items():
        if k in r['Name'] and ('bwd_' + k) not in r['Name']:
            avg = float(r['AverageNs']) / 1e3
            print(f"{k:14s} {int(r['Calls']):4d} calls  avg {avg:7.1f} us  min {float(r['MinNs'])/1e3:7.1f}  -> {t * mb / avg:5.2f} TB/s ({t} maps of {mb:.1f} MB)")
PY
rm -rf gpurun_out/prof_nc
