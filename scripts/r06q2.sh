#!/bin/bash
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p gpurun_out/r06q
python -m pytest tests/test_model_gpu.py -q -s -k "branch3_pool_gradient or 16_bit or bf16 or fp8" 2>&1 | grep -v "^$" | tail -14 > gpurun_out/r06q/t2.txt
cat gpurun_out/r06q/t2.txt
