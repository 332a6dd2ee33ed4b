#!/bin/bash
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p gpurun_out/r06za
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "centred_bf16 or batch_norm or bf16" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py tests/test_frontends_gpu.py -x -q -s -k "bf16 or fp8 or 16_bit or z_storage" 2>&1 | grep -v "^$" | tail -14 | cut -c1-220
