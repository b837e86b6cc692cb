for sh in "8 114 152 128 256 3 2 1" "8 57 76 256 512 3 2 1" "8 28 38 256 512 4 1 1" "8 57 76 384 768 1 1 0" "8 29 38 768 1536 1 1 0" "8 114 152 128 256 1 2 0" "8 58 76 128 768 3 2 1" "8 57 76 128 256 4 2 1"; do
  echo -n "default: "; python tools/conv_bench.py $sh wgrad 30
  echo -n "forced : "; RDFC_WGRAD_GATHER_GLDS=2 python tools/conv_bench.py $sh wgrad 30
done
