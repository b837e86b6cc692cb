#!/bin/bash
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rccl-self-test > gpurun_out/r02_rccl_one_rank_bench.json 2> gpurun_out/r02_rccl.err; tail -5 gpurun_out/r02_rccl.err; head -c 600 gpurun_out/r02_rccl_one_rank_bench.json
