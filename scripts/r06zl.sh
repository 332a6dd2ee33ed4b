#!/bin/bash
# ds_maxpool3_bwd_sums: channel quads per workgroup forced (fewer, larger unit blocks = fewer scattered partial writes)
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
for cw in 0 64 32 16 8; do      # (the first sweep ran with the ">= 10 % more threads" rule as the default)
  echo "== CW <= $cw (0: the default rule)"
  if [ $cw = 0 ]; then python scripts/pool_sums_bench.py 2>/dev/null; else DS_POOL_SUMS_CW=$cw python scripts/pool_sums_bench.py 2>/dev/null; fi
done
