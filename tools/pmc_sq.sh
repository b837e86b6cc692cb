#!/bin/bash
# usage: bash tools/pmc_sq.sh <tag> N H W Ci Co k s p fwd|dgrad|wgrad   -- shader-side counters of one conv shape (tools/conv_bench.py), one set per pass:
# where a wave's cycles go (busy / waiting / issuing), instruction mix, LDS conflicts, MFMA-busy share
tag=$1; shift
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcs_${tag}/p$i -- python3 tools/conv_bench.py "$@" 5 > gpurun_out/pmcs_${tag}_p$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f'gpurun_out/pmcs_{tag}/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        if any(t in k for t in ('igemm', 'patch', 'slab', 'wgrad', 'thin', 'glds', 'convt')):
            tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in tot.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:40s} {v / cnt[(k, c)]:18.1f}  (per launch)')
PY
rm -rf gpurun_out/pmcs_${tag}
