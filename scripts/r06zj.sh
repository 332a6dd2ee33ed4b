#!/bin/bash
# the 16-bit label's scheduling switches re-swept with the round's kernels: side-stream arrangement x block-batched BatchNorm bits,
# B = 256 and cfg5's per-GPU share (B = 128), bf16 and fp8
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2; do
  for sm in 0 1 2; do
    echo "bf16 B256 side=$sm $(run --dtype bf16 --side-mode $sm)"
    echo "bf16 B128 side=$sm $(run --dtype bf16 --batch 128 --side-mode $sm)"
    echo "fp8 B128 side=$sm $(run --dtype fp8 --batch 128 --side-mode $sm)"
  done
  for bb in 0 1 2 3; do
    echo "bf16 B256 batch_bn=$bb $(DS_BATCH_BN=$bb run --dtype bf16)"
    echo "bf16 B128 batch_bn=$bb $(DS_BATCH_BN=$bb run --dtype bf16 --batch 128)"
  done
  for bpc in 8 16 32; do
    echo "bf16 B256 stream_bpc=$bpc $(DS_STREAM_BPC=$bpc run --dtype bf16)"
  done
done | sort
