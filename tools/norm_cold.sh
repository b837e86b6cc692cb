#!/bin/bash
# usage (GPU box): bash tools/norm_cold.sh ["VAR=val VAR2=val" ...]
#   -> per-kernel time of the normalisation kernels on HBM-cold maps of the step's four main shapes; one table per environment
#      setting given (none: the defaults).  RDFC_HIP_LIB=<variant .so> is a setting like any other.
mkdir -p gpurun_out/nc
root=$PWD; cd /tmp && export TMPDIR=/tmp && cd $root
[ $# -eq 0 ] && set -- ""
for setting in "$@"; do
  echo "#### ${setting:-defaults}"
  for shp in "16 228 304 64" "16 114 152 128" "16 57 76 256" "16 29 38 512"; do
    rm -rf gpurun_out/nc/raw
    ( [ -n "$setting" ] && export $setting
      rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nc/raw -- python3 tools/norm_cold.py $shp > /dev/null 2>&1 )
    echo "== $shp"
    python3 - $shp <<'PY'
import csv, glob, sys
N,H,W,C = map(int, sys.argv[1:5])
nbytes = N*H*W*C*2
f = glob.glob('gpurun_out/nc/raw/*/*kernel_stats.csv')[0]
passes = dict(stats_kernel=1, apply_kernel=2, bwd_reduce=2, bwd_apply=3)
for r in csv.DictReader(open(f)):
    n = r['Name']
    for key in ('stats_kernel','bwd_reduce','bwd_apply','apply_kernel','sum_finalize','stats_finalize','chsum'):
        if key in n:
            us = float(r['AverageNs'])/1e3
            tb = f"{passes[key]*nbytes/us/1e6:.2f} TB/s" if key in passes else ''
            print(f"   {key:16s} calls {r['Calls']:>4s} avg {us:7.1f} us   min {float(r['MinNs'])/1e3:7.1f}   {tb}")
            break
PY
  done
done
rm -rf gpurun_out/nc
