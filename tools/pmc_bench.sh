#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_bench.sh r01 [bench flags]
# HBM-side traffic per kernel of the bench command: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (kernel-trace
# only; the two together do not fit the 4 TCC slots), each time-boxed.  Writes gpurun_out/<tag>_pmc_traffic.json:
#   {kernel symbol: {launches, fetch_bytes (already x2: gfx950 tallies 128-B read requests at 64 B), write_bytes}} per launch.
tag=${1:-r01}; shift
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcb_${tag}/$c
  timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcb_${tag}/$c -- python3 bench.py --steps 2 --warmup 1 \
      --no-cpu-baseline --no-kernel-events "$@" > gpurun_out/pmcb_${tag}_$c.log 2>&1
  echo "pass $c rc=$?"
done
python3 - "$tag" "$@" <<'PY'
import csv, glob, sys, collections, json, re
tag = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f'gpurun_out/pmcb_{tag}/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
out = {}
for k, d in tot.items():
    nf, nw = cnt[(k, 'FETCH_SIZE')], cnt[(k, 'WRITE_SIZE')]
    if not nf or not nw: continue
    out[k] = dict(launches=nf, fetch_bytes=2.0 * 1024 * d['FETCH_SIZE'] / nf, write_bytes=1024.0 * d['WRITE_SIZE'] / nw)
sys.path.insert(0, '.')
from rdfc_gan_amd import hip
out['_meta'] = dict(csrc_sha16=hip.build_hash(),        # compiled into the library the passes above loaded command='bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events ' + ' '.join(sys.argv[2:]),
                    note='bytes per launch; FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md); separate --pmc passes')
json.dump(out, open(f'gpurun_out/{tag}_pmc_traffic.json', 'w'), indent=1, sort_keys=True)
for k, v in sorted(((k, v) for k, v in out.items() if k != '_meta'), key=lambda kv: -kv[1]['launches'] * (kv[1]['fetch_bytes'] + kv[1]['write_bytes']))[:14]:
    print(f"{v['launches']:6d} x  rd {v['fetch_bytes']/1e6:8.2f} MB  wr {v['write_bytes']/1e6:8.2f} MB   {k[:110]}")
PY
rm -rf gpurun_out/pmcb_${tag}
