#!/bin/bash
export DS_LIB=${DS_LIB:-$(cd $(dirname $0)/.. && pwd)/tumblr_emotions_amd/libds_kernels_tuning.so}      # the DS_* A/B switches are honoured beside the tuning build only
mkdir -p gpurun_out/r06f
python -m pytest tests/test_dp_gpu.py -x -q -k "bench_self" 2>&1 | tail -40 > gpurun_out/r06f/t0.txt
python -m pytest tests/test_kernels_gpu.py -x -q -k "finalize_inside or branch3 or wide_1x1" 2>&1 | tail -12 > gpurun_out/r06f/t1.txt
python -m pytest tests/test_model_gpu.py -x -q -k "b32_config3 or zcat_step or joint_step" 2>&1 | tail -8 > gpurun_out/r06f/t2.txt
for i in 1 2 3; do for e in 1 0; do
  DS_FUSE_FIN=$e python bench.py --batch 32 --steps 30 --warmup 10 --no-cpu-baseline --no-gather --no-conv-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B32 fuse_fin=$e', d['ms_per_step'])"
done; done > gpurun_out/r06f/b32.txt 2>&1
for e in 1 0; do
  DS_FUSE_FIN=$e python bench.py --batch 64 --steps 30 --warmup 10 --no-cpu-baseline --no-gather --no-conv-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B64 fuse_fin=$e', d['ms_per_step'])"
done >> gpurun_out/r06f/b32.txt 2>&1
cat gpurun_out/r06f/*.txt
