run() { python bench.py --steps 16 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"; }
echo -n "default            : "; run
echo -n "RDFC_GLDS=0        : "; RDFC_GLDS=0 run
echo -n "RDFC_WGRAD_GLDS=0  : "; RDFC_WGRAD_GLDS=0 run
echo -n "IGEMM_GLDS off     : "; RDFC_IGEMM_GLDS_MINSTEPS=100000 run
echo -n "single stream      : "; RDFC_BRANCH_STREAMS=0 RDFC_NET_STREAMS=0 RDFC_WGRAD_STREAM=0 run
echo -n "default            : "; run
