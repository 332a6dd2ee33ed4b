#!/bin/bash
export DS_LIB=${DS_LIB:-$(cd $(dirname $0)/.. && pwd)/tumblr_emotions_amd/libds_kernels_tuning.so}      # the DS_* A/B switches are honoured beside the tuning build only
mkdir -p gpurun_out/r06j
python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -6 > gpurun_out/r06j/t1.txt
for i in 1 2 3; do for e in 1 0; do
  DS_STEM_SUMS=$e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 stem_sums=$e', d['ms_per_step'])"
done; done > gpurun_out/r06j/ab.txt 2>&1
for e in 1 0; do
  DS_STEM_SUMS=$e python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 stem_sums=$e', d['ms_per_step'])"
done >> gpurun_out/r06j/ab.txt 2>&1
cat gpurun_out/r06j/*.txt
