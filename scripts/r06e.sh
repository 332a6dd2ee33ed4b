#!/bin/bash
mkdir -p gpurun_out/r06e
python -m pytest tests/test_kernels_gpu.py -x -q -k "stem" 2>&1 | tail -5 > gpurun_out/r06e/t1.txt
python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py -x -q -k "stem_with or bf16 or fp8" 2>&1 | tail -8 > gpurun_out/r06e/t2.txt
python scripts/stem_bench.py 256 2>&1 | tail -4 > gpurun_out/r06e/stem_bench.txt
for i in 1 2 3; do for f in "" "--no-stem-pool"; do
  python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing $f 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 $f', d['ms_per_step'])"
done; done > gpurun_out/r06e/ab_bf16.txt 2>&1
for f in "" "--no-stem-pool"; do
  python bench.py --dtype fp8 --batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing $f 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp8 B128 $f', d['ms_per_step'])"
done >> gpurun_out/r06e/ab_bf16.txt 2>&1
cat gpurun_out/r06e/*.txt
