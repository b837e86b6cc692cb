#!/bin/bash
# usage (GPU box, repo root): bash tools/final_round.sh r03   -> every artefact of a round under gpurun_out/<tag>_*
tag=${1:-r06}
mkdir -p gpurun_out
bash tools/pmc_bench.sh $tag                                                # HBM-side bytes per kernel (separate --pmc passes)
cp gpurun_out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json       # (this box's copy: the bench line below cites counters of THIS build)
# round 6: the same three passes for the extra legs of the default bench line (bench.py pmc_table: suffix per workload), so that
# as_worded_bf16 / configs.#3 / configs.#5 carry roofline.traffic and mfma_busy too
PMC_SUFFIX=_bf16 bash tools/pmc_bench.sh $tag --dtype bf16
PMC_SUFFIX=_nlspn bash tools/pmc_bench.sh $tag --nlspn
PMC_SUFFIX=_480x640 bash tools/pmc_bench.sh $tag --height 480 --width 640 --batch 4
for sfx in _bf16 _nlspn _480x640; do cp gpurun_out/${tag}_pmc_traffic$sfx.json profiles/${tag}_pmc_traffic$sfx.json; done
bash tools/profile_round.sh $tag                                            # default bench (with CPU baseline) + the same command under rocprofv3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-as-worded --rccl-self-test > gpurun_out/${tag}_rccl_one_rank_bench.json 2> gpurun_out/${tag}_rccl.err
bash tools/timeline.sh --no-as-worded > gpurun_out/${tag}_timeline.txt 2>&1
bash tools/one_stream_stats.sh ${tag} --no-as-worded > /dev/null 2>&1      # every kernel alone on the chip (one stream): gpurun_out/<tag>_one_stream_kernel_stats.csv
# BASELINE config #3 (NLSPN on): bench line + kernel stats
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-as-worded --nlspn > gpurun_out/${tag}_nlspn_bench.json 2>/dev/null
root=$PWD; cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/prof_${tag}n
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}n -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-as-worded --nlspn > /dev/null 2>&1
cp gpurun_out/prof_${tag}n/*/*_kernel_stats.csv gpurun_out/${tag}_nlspn_kernel_stats.csv; rm -rf gpurun_out/prof_${tag}n
# the same workload in bf16 (BASELINE configs[1] as worded; the default is f16), config #5's per-GPU shape (fp16, 480x640, batch 4) and the f32 parity mode
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dtype bf16 > gpurun_out/${tag}_bf16_bench.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype f16 --height 480 --width 640 --batch 4 > gpurun_out/${tag}_f16_480x640_bench.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dtype f32 > gpurun_out/${tag}_f32_bench.json 2>/dev/null
# every conv shape alone (one stream)
RDFC_BRANCH_STREAMS=0 RDFC_NET_STREAMS=0 RDFC_WGRAD_STREAM=0 RDFC_D_UNDER_G=0 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-as-worded --event-every 1 --by-shape gpurun_out/${tag}_by_shape_one_stream.txt > gpurun_out/${tag}_bench_one_stream.json 2>/dev/null
JRS=4,6,8,10 python tools/conv_tall_bench.py 3 10 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_conv_tall_ab.txt     # the 3x3 kernel at every forced tile height against the plan's choice
# round 6: the narrow-map launches and the epilogue addends alone, the NLSPN chains alone, host slack, 1 000-step soaks in f16 and in the f32 parity mode
bash tools/narrow_bench.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_narrow_bench.txt
python tools/addend_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_addend_bench.txt
python tools/nlspn_bench.py 16 228 304 18 1.5 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_nlspn_chains.txt
python tools/host_slack.py 20 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_host_slack.txt
SOAK_EVERY=100 python tools/soak.py 1000 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_soak_f16.txt
SOAK_DTYPE=f32 SOAK_EVERY=100 python tools/soak.py 1000 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_soak_f32.txt
for f in ${tag}_bench ${tag}_bench_under_rocprof ${tag}_rccl_one_rank_bench ${tag}_nlspn_bench ${tag}_bf16_bench ${tag}_f16_480x640_bench ${tag}_f32_bench ${tag}_bench_one_stream; do
python3 - gpurun_out/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], 'pairs/s', d['ms_per_step'], 'ms', d.get('roofline',{}).get('frac'), d.get('cpu_baseline',{}).get('value'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
head -30 gpurun_out/${tag}_by_shape_one_stream.txt
