for cfg in "8 512" "4 512" "16 512" "8 1024" "8 256" "32 512"; do
  set -- $cfg
  echo "== min_rows $1 want_wg $2"
  for sh in "8 57 76 256" "8 114 152 128" "8 29 38 512"; do
    RDFC_NORM_MIN_ROWS=$1 RDFC_NORM_WANT_WG=$2 bash tools/norm_cold.sh $sh 2>/dev/null | grep -E "bwd_reduce|stats_kernel" | awk -v s="$sh" '{print "   " s " : " $1, $5, $6}'
  done
done
