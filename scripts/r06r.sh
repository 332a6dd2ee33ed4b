#!/bin/bash
# the six wgrad launches of the headline step (four Mixed_5c convs, two LSTM matrices): tile shapes and prefetch depth
R=$(cd $(dirname $0)/.. && pwd)
T=$R/tumblr_emotions_amd/libds_kernels_tuning.so
U8=$R/tumblr_emotions_amd/csrc/build_tuning/libds_tuning_u8.so
mkdir -p gpurun_out/r06r
export WGRAD_ONLY=0,1,2,3,4,5
{
echo "== planner's choice (U = 4)"; DS_LIB=$T DS_WGRAD_DEBUG=1 python scripts/wgrad_bench.py 2>&1 | grep -v amdgpu.ids
for f in 1,4 2,4 4,2 2,2 4,4 4,1; do
  echo "== forced $f U = 4"; DS_LIB=$T DS_WGRAD_FORCE=$f,0 python scripts/wgrad_bench.py 2>&1 | grep -v "amdgpu.ids\|overrides"
  echo "== forced $f U = 8"; DS_LIB=$U8 DS_WGRAD_FORCE=$f,0 python scripts/wgrad_bench.py 2>&1 | grep -v "amdgpu.ids\|overrides"
done
} > gpurun_out/r06r/wgrad.txt 2>&1
cat gpurun_out/r06r/wgrad.txt
