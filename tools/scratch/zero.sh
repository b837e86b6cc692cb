for sh in "16 57 76 256 256 3 1 1" "8 57 76 256 256 3 1 1"; do
  for m in 1 6 4; do
    echo -n "mode $m random: "; RDFC_GLDS=$m python tools/conv_bench.py $sh fwd 50
    echo -n "mode $m zeros : "; RDFC_BENCH_ZERO=1 RDFC_GLDS=$m python tools/conv_bench.py $sh fwd 50
  done
done
