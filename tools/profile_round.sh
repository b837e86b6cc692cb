#!/bin/bash
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r01 [bench flags]
# Runs bench.py plain (-> gpurun_out/<tag>_bench.json) and the same command under rocprofv3 --kernel-trace --stats
# (-> gpurun_out/<tag>_kernel_stats.csv + the bench line printed by the profiled run).
tag=${1:-r01}; shift
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 3 "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 600 gpurun_out/${tag}_bench.err
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/prof_${tag}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-as-worded "$@" \
    > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_rocprof.err
cp gpurun_out/prof_${tag}/*/*_kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_${tag}
head -c 1500 gpurun_out/${tag}_bench.json
