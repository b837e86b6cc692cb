mkdir -p gpurun_out/r2b
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 10 --warmup 3 --graph --no-cpu-baseline --no-kernel-events > gpurun_out/r2b/$tag.json 2> gpurun_out/r2b/$tag.err; echo "$tag rc=$? $(tail -c 400 gpurun_out/r2b/$tag.json | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["launch"])
except Exception as e: print("ERR", e)')"; }
run branch_only RDFC_NET_STREAMS=0 RDFC_WGRAD_STREAM=0
run wgrad_only RDFC_NET_STREAMS=0 RDFC_BRANCH_STREAMS=0
run net_only RDFC_BRANCH_STREAMS=0 RDFC_WGRAD_STREAM=0
run branch_wgrad RDFC_NET_STREAMS=0
