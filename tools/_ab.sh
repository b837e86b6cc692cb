for rep in 1 2; do for e in "A=1" "RDFC_GLDS_STACK=0"; do
echo "$e: $(env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --height 480 --width 640 --batch 4 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done; done
