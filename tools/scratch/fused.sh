run() { python bench.py --steps 20 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"; }
for rep in 1 2; do
echo -n "default       : "; run
echo -n "fused stats   : "; RDFC_FUSED_STATS=1 run
done
