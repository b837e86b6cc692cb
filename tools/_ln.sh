python bench.py --steps 10 --warmup 3 --no-cpu-baseline --label-net 2>/dev/null | tail -1 | cut -c1-330
RDFC_BRANCH_STREAMS=0 RDFC_NET_STREAMS=0 RDFC_WGRAD_STREAM=0 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --label-net --event-every 1 --by-shape gpurun_out/ln_by_shape.txt 2>/dev/null | tail -1 | cut -c1-200
