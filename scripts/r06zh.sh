#!/bin/bash
# fp8 label: z16 also for the layers whose forward runs on ds_conv_fp8
R=$(cd $(dirname $0)/.. && pwd)
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -x -q -s -k "fp8 or z_storage or 16_bit" 2>&1 | grep -v "^$" | tail -12 | cut -c1-230
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do echo "fp8 $(run --dtype fp8)"; echo "fp8_B128 $(run --dtype fp8 --batch 128)"; echo "bf16 $(run --dtype bf16)"; done
