for n in 1 2 3 4; do python tools/conv_bench.py $n 57 76 256 256 3 1 1 fwd 30; done
echo "one workgroup per CU:"
for n in 2 4 8 16; do RDFC_GLDS_PAD_LDS=49152 python tools/conv_bench.py $n 57 76 256 256 3 1 1 fwd 30; done
