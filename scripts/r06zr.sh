#!/bin/bash
# 16-bit labels: the 3x3 input gradients of the two 28 x 28 layers with >= 128 reduction channels on the direct kernel (a ds_conv_plan rule) and bf16 dz
# for the 3x3 layers whose dgrad runs there.  Measured: nothing at the step level; the rule and the switch were removed again (r06_notes)
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "bf16 B256 old $(DS_PLAN_DIRECT28=0 DS_DZ16_DIRECT3=0 run --dtype bf16)"; echo "bf16 B256 new $(run --dtype bf16)"
  echo "bf16 B256 old-rule+dz16-14x14 $(DS_PLAN_DIRECT28=0 run --dtype bf16)"
  echo "bf16 B128 old $(DS_PLAN_DIRECT28=0 DS_DZ16_DIRECT3=0 run --dtype bf16 --batch 128)"; echo "bf16 B128 new $(run --dtype bf16 --batch 128)"
  echo "bf16 B128 old-rule+dz16-14x14 $(DS_PLAN_DIRECT28=0 run --dtype bf16 --batch 128)"
  echo "fp8 B256 old $(DS_PLAN_DIRECT28=0 DS_DZ16_DIRECT3=0 run --dtype fp8)"; echo "fp8 B256 new $(run --dtype fp8)"
done | sort
