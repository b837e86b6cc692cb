#!/bin/bash
# per-kernel times of the two weight-gradient paths (rocprofv3 kernel trace over tools/wgrad_ring_bench.py, few iterations)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wr_trace -- python3 $R/tools/wgrad_ring_bench.py 2 5 > $R/gpurun_out/wr_trace.log 2>&1
f=$(ls $R/gpurun_out/wr_trace/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/wr_kernel_stats.csv
