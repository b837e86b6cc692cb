#!/bin/bash
# usage (GPU box): bash tools/glds_stamps.sh N H W Ci Co [tile [fwd|dgrad] [JR]]   -- builds the stamped variant and prints the per-phase breakdown (tile: the per-tile timeline, tools/glds_tile_stamps.py)
shape="$1 $2 $3 $4 $5"; shift 5
mode=phase; if [ "$1" = tile ]; then mode=tile; shift; fi
bash tools/build_variant.sh conv_glds.hip variants/stamps.so -DRDFC_GLDS_STAMPS || exit 1
if [ $mode = tile ]; then RDFC_HIP_LIB=variants/stamps.so python tools/glds_tile_stamps.py $shape "$@"; else RDFC_HIP_LIB=variants/stamps.so python tools/glds_stamps.py $shape; fi
