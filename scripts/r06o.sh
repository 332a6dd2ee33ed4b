#!/bin/bash
# the BatchNorm launches of a block's three closing layers once per block (InceptionV1Engine.batch_bn)
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=${DS_LIB:-$R/tumblr_emotions_amd/libds_kernels_tuning.so}
mkdir -p gpurun_out/r06o
python -m pytest tests/test_kernels_gpu.py -x -q -k "three_layers_as_one or batch_norm" 2>&1 | tail -6 > gpurun_out/r06o/t1.txt
python -m pytest tests/test_model_gpu.py -x -q -k "batched_batch_norm or zcat or branch3 or stem_with or oracle" 2>&1 | tail -6 > gpurun_out/r06o/t2.txt
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do for e in 1 0; do echo "B256 batch_bn=$e $(DS_BATCH_BN=$e run)"; done; done > gpurun_out/r06o/ab.txt 2>&1
for i in 1 2 3; do for e in 1 0; do echo "B32 batch_bn=$e $(DS_BATCH_BN=$e run --batch 32)"; done; done >> gpurun_out/r06o/ab.txt 2>&1
for e in 1 0; do echo "B64 batch_bn=$e $(DS_BATCH_BN=$e run --batch 64)"; echo "B128 batch_bn=$e $(DS_BATCH_BN=$e run --batch 128)"; echo "image B128 batch_bn=$e $(DS_BATCH_BN=$e run --batch 128 --mode image)"; done >> gpurun_out/r06o/ab.txt 2>&1
unset DS_LIB
bash scripts/step_dispatches.sh r06o > /dev/null 2>&1
head -3 gpurun_out/r06o_step_dispatches.txt > gpurun_out/r06o/disp.txt
cat gpurun_out/r06o/*.txt
