#!/bin/bash
mkdir -p gpurun_out/r06d
python -m pytest tests/test_kernels_gpu.py -x -q -k "stem" 2>&1 | tail -5 > gpurun_out/r06d/t1.txt
python -m pytest tests/test_model_gpu.py -x -q -k "stem_with" 2>&1 | tail -5 > gpurun_out/r06d/t2.txt
python scripts/stem_bench.py 256 2>&1 | tail -1 > gpurun_out/r06d/stem_bench.txt
python scripts/stem_bench.py 32 2>&1 | tail -1 >> gpurun_out/r06d/stem_bench.txt
bash scripts/ab.sh --no-stem-pool 3 > gpurun_out/r06d/ab_stem.txt 2>&1
cat gpurun_out/r06d/*.txt
