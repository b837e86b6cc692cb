#!/bin/bash
# usage: bash tools/pmc_traffic.sh <tag> N H W Ci Co k s p fwd|dgrad|wgrad
# HBM-side traffic of one conv shape (FETCH_SIZE / WRITE_SIZE, own pass, kernel-trace only); see MI355X_MICROARCH.md
# 'HBM': on gfx950 FETCH_SIZE counts wide coalesced reads at half their bytes -> doubled here.
tag=$1; shift
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/pmc_${tag}
# one counter per pass: FETCH_SIZE takes 3 of the 4 TCC slots and WRITE_SIZE 2, both in one pass do not fit (and hung the run)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${tag}/$c -- python3 tools/conv_bench.py "$@" 5 > gpurun_out/pmc_${tag}_$c.log 2>&1
  echo "pass $c rc=$?"
done
python3 - "$tag" "$@" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f'gpurun_out/pmc_{tag}/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:80]
        if any(t in k for t in ('igemm', 'patch', 'slab', 'wgrad', 'thin', 'glds')):
            tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in tot.items():
    f = d['FETCH_SIZE'] / cnt[(k, 'FETCH_SIZE')]; w = d['WRITE_SIZE'] / cnt[(k, 'WRITE_SIZE')]
    print(f'{k}\n   FETCH_SIZE {f:.0f} KB (x2 = {2*f/1e3:.1f} MB)  WRITE_SIZE {w:.0f} KB ({w/1e3:.1f} MB)  per launch  shape {" ".join(sys.argv[2:])}')
PY
