#!/bin/bash
# round 6, second GPU call: the stem with MaxPool_2a inside -- kernel + step tests, interleaved A/B, stem microbench
mkdir -p gpurun_out/r06b
python -m pytest tests/test_kernels_gpu.py -x -q -k "stem or branch3" 2>&1 | tail -15 > gpurun_out/r06b/t1.txt
python -m pytest tests/test_model_gpu.py -x -q -k "stem_with or branch3 or zcat_step or joint" 2>&1 | tail -15 > gpurun_out/r06b/t2.txt
bash scripts/ab.sh --no-stem-pool 3 > gpurun_out/r06b/ab_stem.txt 2>&1
for b in 32 128; do for f in "" "--no-stem-pool"; do
  python bench.py --batch $b --steps 30 --warmup 10 --no-cpu-baseline --no-gather --no-conv-timing $f 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B$b $f', d['ms_per_step'])"
done; done > gpurun_out/r06b/b32.txt 2>&1
cat gpurun_out/r06b/*.txt
