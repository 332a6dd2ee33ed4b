#!/bin/bash
# round 6, first GPU call: the fused Branch_3 (pool on load) -- kernel + step tests, interleaved A/B at B = 256, B = 32
mkdir -p gpurun_out/r06a
python -m pytest tests/test_kernels_gpu.py -x -q -k "branch3 or wide_1x1 or applied_on_load" 2>&1 | tail -15 > gpurun_out/r06a/t1.txt
python -m pytest tests/test_model_gpu.py -x -q -k "branch3 or zcat_step or b32_config3" 2>&1 | tail -15 > gpurun_out/r06a/t2.txt
bash scripts/ab.sh --no-fuse-b3 3 > gpurun_out/r06a/ab_b3.txt 2>&1
for b in 32; do for f in "" "--no-fuse-b3"; do
  python bench.py --batch $b --steps 30 --warmup 10 --no-cpu-baseline --no-gather --no-conv-timing $f 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B$b $f', d['ms_per_step'])"
done; done > gpurun_out/r06a/b32.txt 2>&1
cat gpurun_out/r06a/*.txt
