python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for shp in "16 228 304 64 64 3 1 1" "16 57 76 256 256 3 1 1" "16 114 152 128 128 3 1 1" "16 29 38 512 512 3 1 1"; do
  python tools/conv_bench.py $shp fwd 30 2>/dev/null | tail -1
  RDFC_GLDS_ROWSTORE=0 python tools/conv_bench.py $shp fwd 30 2>/dev/null | tail -1 | sed 's/^/   direct: /'
done; done
b() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-as-worded 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"; }
b rows; RDFC_GLDS_ROWSTORE=0 b direct; b rows; RDFC_GLDS_ROWSTORE=0 b direct
