#!/bin/bash
# ds_maxpool3_bwd_sums with a workgroup per CHUNK of channel quads (widths whose quads do not divide 256) against one chunk
R=$(cd $(dirname $0)/.. && pwd)
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "max_pool_gradient" 2>&1 | tail -3
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
echo "== one chunk"; DS_POOL_SUMS_CHUNKS=0 python scripts/pool_sums_bench.py 2>/dev/null
echo "== chunks";    python scripts/pool_sums_bench.py 2>/dev/null
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "bf16 B256 one_chunk $(DS_POOL_SUMS_CHUNKS=0 run --dtype bf16)"; echo "bf16 B256 chunks $(run --dtype bf16)"
  echo "bf16 B128 one_chunk $(DS_POOL_SUMS_CHUNKS=0 run --dtype bf16 --batch 128)"; echo "bf16 B128 chunks $(run --dtype bf16 --batch 128)"
  echo "fp8 B256 one_chunk $(DS_POOL_SUMS_CHUNKS=0 run --dtype fp8)"; echo "fp8 B256 chunks $(run --dtype fp8)"
done | sort
