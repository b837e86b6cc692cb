#!/bin/bash
# usage (GPU box): bash tools/glds_stamps.sh N H W Ci Co [extra -D flags]   -- builds the stamped variant and prints the breakdown
shape="$1 $2 $3 $4 $5"; shift 5
bash tools/build_variant.sh conv_glds.hip gpurun_out/variants/stamps.so -DRDFC_GLDS_STAMPS "$@" || exit 1
RDFC_HIP_LIB=gpurun_out/variants/stamps.so python tools/glds_stamps.py $shape
