#!/bin/bash
# the 16-bit labels' reference tests with z in bf16 storage forced on (DS_Z16=1 beside the tuning library)
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
export DS_Z16=1
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py -q -s -k "bf16 or fp8 or 16_bit" 2>&1 | grep -v "^$" | tail -30 | cut -c1-250
