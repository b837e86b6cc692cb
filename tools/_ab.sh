python -m pytest tests/test_kernels_gpu.py -x -q -k "test_conv_fwd_dgrad_wgrad" 2>&1 | tail -3
RDFC_BRANCH_STREAMS=0 RDFC_NET_STREAMS=0 RDFC_WGRAD_STREAM=0 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-as-worded --event-every 1 --by-shape gpurun_out/r04a_by_shape.txt > /dev/null 2>&1
grep -E "thin" gpurun_out/r04a_by_shape.txt
