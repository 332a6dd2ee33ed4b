#!/bin/bash
export DS_LIB=${DS_LIB:-$(cd $(dirname $0)/.. && pwd)/tumblr_emotions_amd/libds_kernels_tuning.so}      # the DS_* A/B switches are honoured beside the tuning build only
mkdir -p gpurun_out/r06g
python -m pytest tests/test_model_gpu.py -x -q -k "bf16_dz or 16_bit_dgrad or bf16" 2>&1 | tail -8 > gpurun_out/r06g/t1.txt
for i in 1 2 3; do for e in 1 0; do
  DS_DZ16=$e python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 dz16=$e', d['ms_per_step'])"
done; done > gpurun_out/r06g/ab.txt 2>&1
for e in 1 0; do
  DS_DZ16=$e python bench.py --dtype fp8 --batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp8 B128 dz16=$e', d['ms_per_step'])"
done >> gpurun_out/r06g/ab.txt 2>&1
cat gpurun_out/r06g/*.txt
