mkdir -p gpurun_out/r2o
root=$PWD; cd /tmp && export TMPDIR=/tmp && cd $root
for shp in "16 228 304 64 2" "16 114 152 128 2" "16 57 76 256 2" "16 29 38 512 2"; do
  tag=$(echo $shp | tr ' ' '_')
  rm -rf gpurun_out/r2o/raw
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2o/raw -- python3 tools/norm_bench.py $shp > gpurun_out/r2o/out_$tag.txt 2>/dev/null
  echo "== $shp"; cat gpurun_out/r2o/out_$tag.txt | grep -v amdgpu
  python3 - $shp <<'PY'
import csv, glob, sys
N,H,W,C,G = map(int, sys.argv[1:6])
nbytes = N*H*W*C*2
f = glob.glob('gpurun_out/r2o/raw/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    for key in ('stats_kernel','apply_kernel','bwd_reduce','bwd_apply','sum_finalize','stats_finalize','chsum'):
        if key in n:
            us = float(r['AverageNs'])/1e3
            print(f"   {key:16s} calls {r['Calls']:>4s} avg {us:7.1f} us   min {float(r['MinNs'])/1e3:7.1f}   ({nbytes/1e6:.1f} MB map -> {nbytes/us/1e6:.2f} TB/s per map-pass)")
            break
PY
done
rm -rf gpurun_out/r2o/raw
