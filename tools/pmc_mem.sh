#!/bin/bash
# usage: bash tools/pmc_mem.sh <tag> N H W Ci Co k s p fwd|dgrad|wgrad   -- memory-path counters (TA / TCP / TCC / GRBM), one set per pass
tag=$1; shift
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
i=0
# (a TA_* set -- TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum ... -- did not finish within the time box on this pool and is left out)
for set in "TCC_BUSY_avr TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_TA_BUSY" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcm_${tag}/p$i -- python3 tools/conv_bench.py "$@" 5 > gpurun_out/pmcm_${tag}_p$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f'gpurun_out/pmcm_{tag}/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        if any(t in k for t in ('igemm', 'patch', 'slab', 'wgrad', 'thin', 'glds')):
            tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in tot.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:40s} {v / cnt[(k, c)]:18.1f}  (per launch)')
PY
rm -rf gpurun_out/pmcm_${tag}
