tag=r04
bash tools/one_stream_stats.sh ${tag} --no-as-worded > /dev/null 2>&1
RDFC_BRANCH_STREAMS=0 RDFC_NET_STREAMS=0 RDFC_WGRAD_STREAM=0 RDFC_D_UNDER_G=0 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-as-worded --event-every 1 --by-shape gpurun_out/${tag}_by_shape_one_stream.txt > gpurun_out/${tag}_bench_one_stream.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_one_stream.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
d=json.loads(open('gpurun_out/r04_one_stream_under_rocprof.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])
PY
head -8 gpurun_out/r04_by_shape_one_stream.txt
