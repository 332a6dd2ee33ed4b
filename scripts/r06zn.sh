#!/bin/bash
# wgrad tile shapes forced on all fourteen shapes of wgrad_bench.py (the six of the headline step + a sample of train_all's)
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
echo "== planner"; python scripts/wgrad_bench.py 2>&1 | grep -v "amdgpu.ids\|overrides"
for f in 2,4 2,2 4,4 4,2; do
  echo "== forced $f"; DS_WGRAD_FORCE=$f,0 python scripts/wgrad_bench.py 2>&1 | grep -v "amdgpu.ids\|overrides"
done
