#!/bin/bash
# usage (GPU box, repo root): [PMC_SUFFIX=_bf16] bash tools/pmc_bench.sh r01 [bench flags]
# PMC_SUFFIX names the workload when the flags select another one than the headline (bench.py pmc_table: '', _bf16, _nlspn, _480x640):
# the summary goes to gpurun_out/<tag>_pmc_traffic<suffix>.json.
# HBM-side traffic per kernel of the bench command: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (kernel-trace
# only; the two together do not fit the 4 TCC slots), each time-boxed.  Writes gpurun_out/<tag>_pmc_traffic.json:
#   {kernel symbol: {launches, fetch_bytes (already x2: gfx950 tallies 128-B read requests at 64 B), write_bytes}} per launch.
tag=${1:-r01}; shift
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
# third pass: the matrix pipe's busy cycles beside the shader's busy cycles per kernel (north star: "MFMA-busy against gfx950 peak")
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA"; do
  name=$(echo $c | cut -d' ' -f1)
  rm -rf gpurun_out/pmcb_${tag}/$name
  timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcb_${tag}/$name -- python3 bench.py --steps 2 --warmup 1 \
      --no-cpu-baseline --no-kernel-events --no-as-worded "$@" > gpurun_out/pmcb_${tag}_$name.log 2>&1
  echo "pass $name rc=$?"
done
python3 - "$tag" "$@" <<'PY'
import csv, glob, os, sys, collections, json, re
tag = sys.argv[1]
suffix = os.environ.get('PMC_SUFFIX', '')
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
    nm = cnt[(k, 'SQ_VALU_MFMA_BUSY_CYCLES')]
    if nm and d.get('SQ_BUSY_CYCLES'):
        # SQ_VALU_MFMA_BUSY_CYCLES: cycles the matrix pipes are busy, summed over the SIMDs that report (MI355X_MICROARCH.md: = 32 per
        # 32x32x16 MFMA); SQ_BUSY_CYCLES: cycles the shader engines are busy, summed over their instances.  The ratio is comparable
        # between kernels of one profile, not an absolute utilisation (no gfx950 section in the derived-counter tables).
        out[k].update(mfma_busy_cycles=d['SQ_VALU_MFMA_BUSY_CYCLES'] / nm, sq_busy_cycles=d['SQ_BUSY_CYCLES'] / nm,
                      sq_wave_cycles=d['SQ_WAVE_CYCLES'] / nm, mfma_insts=d['SQ_INSTS_MFMA'] / nm,
                      mfma_busy_per_sq_busy=d['SQ_VALU_MFMA_BUSY_CYCLES'] / d['SQ_BUSY_CYCLES'])
sys.path.insert(0, '.')
from rdfc_gan_amd import hip
# (the hash is the one compiled into the library the passes above loaded)
# fabric bytes of a step: every kernel's (FETCH_SIZE x 2 + WRITE_SIZE) x launches over the profiled process / the optimisation steps it ran
# (two adam launches -- G and D -- per step; the few forward-only kernels of bench.py's untimed rmse_dev_m passes ride along: < 1 %)
nsteps = max(1, sum(v['launches'] for k, v in out.items() if 'adam' in k and 'kernel' in k) // 2)
CATS = (('normalisation', ('bwd_apply', 'bwd_reduce', 'apply_kernel', 'stats_kernel', 'finalize', 'chsum', 'adain', 'norm_', 'bn_')),
        ('weight gradients', ('wgrad',)), ('slab_reduce', ('slab_reduce',)),
        ('forward / input-gradient contractions', ('conv3x3', 'igemm', 'convt', 'conv_s2', 'gemm1x1', 'thin_', 'head_dgrad', 'conv_patch')),
        ('nlspn', ('prop_', 'affinity', 'mdconv')),
        ('optimizer / packing', ('adam', 'pack_', 'nonfinite', 'loss_scale')))
cat = collections.defaultdict(float)
for k, v in out.items():
    b = v['launches'] * (v['fetch_bytes'] + v['write_bytes'])
    name = k.lower()
    for c, keys in CATS:
        if any(x in name for x in keys):
            cat[c] += b
            break
    else:
        cat['glue passes (copies, sums, activations, losses, ...)'] += b
out['_meta'] = dict(csrc_sha16=hip.build_hash(), steps_profiled=nsteps,
                    fabric_gb_per_step=round(sum(cat.values()) / nsteps / 1e9, 2),
                    fabric_gb_by_category={c: round(b / nsteps / 1e9, 2) for c, b in sorted(cat.items(), key=lambda kv: -kv[1])},
                    command='bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-as-worded ' + ' '.join(sys.argv[2:]),
                    note='bytes per launch; FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md); separate --pmc passes')
json.dump(out, open(f'gpurun_out/{tag}_pmc_traffic{suffix}.json', 'w'), indent=1, sort_keys=True)
print('fabric GB per step:', out['_meta']['fabric_gb_per_step'], out['_meta']['fabric_gb_by_category'], 'steps', nsteps)
for k, v in sorted(((k, v) for k, v in out.items() if k != '_meta'), key=lambda kv: -kv[1]['launches'] * (kv[1]['fetch_bytes'] + kv[1]['write_bytes']))[:14]:
    print(f"{v['launches']:6d} x  rd {v['fetch_bytes']/1e6:8.2f} MB  wr {v['write_bytes']/1e6:8.2f} MB   {k[:110]}")
PY
rm -rf gpurun_out/pmcb_${tag}
