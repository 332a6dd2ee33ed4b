#!/bin/bash
# wgrad: the two LSTM matrices (2048 output columns) on 64 / 128-row tiles (DS_WGRAD_TALL=0: the planner's 32-row tile).  Measured, not kept: the rule was removed again (r06_notes)
R=$(cd $(dirname $0)/.. && pwd)
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad" 2>&1 | tail -3
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
export WGRAD_ONLY=0,1,2,3,4,5
echo "== 32-row tiles"; DS_WGRAD_TALL=0 python scripts/wgrad_bench.py 2>&1 | grep -v "amdgpu.ids\|overrides"
echo "== rule"; DS_WGRAD_DEBUG=1 python scripts/wgrad_bench.py 2>&1 | grep -v "amdgpu.ids\|overrides" | sort -u
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "f32 B256 tall0 $(DS_WGRAD_TALL=0 run)"; echo "f32 B256 tall1 $(run)"
  echo "bf16 B256 tall0 $(DS_WGRAD_TALL=0 run --dtype bf16)"; echo "bf16 B256 tall1 $(run --dtype bf16)"
  echo "f32 B128 tall0 $(DS_WGRAD_TALL=0 run --batch 128)"; echo "f32 B128 tall1 $(run --batch 128)"
  echo "f32 B32 tall0 $(DS_WGRAD_TALL=0 run --batch 32)"; echo "f32 B32 tall1 $(run --batch 32)"
  echo "text B256 tall0 $(DS_WGRAD_TALL=0 run --mode text)"; echo "text B256 tall1 $(run --mode text)"
done | sort
