for sh in "8 114 152 192 384 1 1 0" "8 57 76 384 768 1 1 0" "8 29 38 768 1536 1 1 0" "8 228 304 64 128 3 2 1" "8 114 152 128 256 1 2 0" "8 228 304 96 192 1 1 0"; do
  for w in fwd dgrad; do
  echo -n "min32: "; python tools/conv_bench.py $sh $w 30
  echo -n "min4 : "; RDFC_IGEMM_GLDS_MINSTEPS=4 python tools/conv_bench.py $sh $w 30
  done
done
