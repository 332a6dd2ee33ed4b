#!/bin/bash
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
mkdir -p gpurun_out/r06v
{
for B in 32 64; do
echo "== model's choice"; python scripts/splitk_bench.py $B 2>&1 | grep -v "amdgpu.ids\|overrides"
for s in 2 3 4; do echo "== forced $s slices"; DS_WINO4_SPLITK=$s python scripts/splitk_bench.py $B 2>&1 | grep -v "amdgpu.ids\|overrides"; done
done
} > gpurun_out/r06v/layers.txt 2>&1
cat gpurun_out/r06v/layers.txt
