#!/bin/bash
# eight row groups per workgroup in the persistent LSTM kernels (the joint model's setting from round 6 on) against four
R=$(cd $(dirname $0)/.. && pwd)
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_frontends_gpu.py -x -q -k "lstm or text or joint" 2>&1 | tail -3
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do echo "f32 rows=4 $(run --lstm-rows 4)"; echo "f32 rows=8 $(run)"; echo "bf16 rows=4 $(run --dtype bf16 --lstm-rows 4)"; echo "bf16 rows=8 $(run --dtype bf16)"; echo "B128 default $(run --batch 128)"; done | sort
