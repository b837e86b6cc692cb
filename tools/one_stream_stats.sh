#!/bin/bash
# usage (GPU box, repo root): bash tools/one_stream_stats.sh <tag> [bench flags]
# one-stream step under rocprofv3 --kernel-trace --stats: with one stream no two kernels overlap, so every kernel's duration is
# its duration ALONE on the chip -> gpurun_out/<tag>_one_stream_kernel_stats.csv (the sum is the one-stream step)
tag=${1:-r04}; shift
mkdir -p gpurun_out
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/prof_${tag}_1s
RDFC_BRANCH_STREAMS=0 RDFC_NET_STREAMS=0 RDFC_WGRAD_STREAM=0 RDFC_D_UNDER_G=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_1s -- \
    python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-as-worded "$@" > gpurun_out/${tag}_one_stream_under_rocprof.json 2> gpurun_out/${tag}_1s.err
cp gpurun_out/prof_${tag}_1s/*/*_kernel_stats.csv gpurun_out/${tag}_one_stream_kernel_stats.csv
rm -rf gpurun_out/prof_${tag}_1s
tail -c 400 gpurun_out/${tag}_one_stream_under_rocprof.json
