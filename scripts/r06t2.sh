#!/bin/bash
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=${DS_LIB:-$R/tumblr_emotions_amd/libds_kernels_tuning.so}
mkdir -p gpurun_out/r06t
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "as_one_launch" 2>&1 | tail -2
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2; do for e in 1 0; do echo "B256 fin_apply=$e $(DS_FIN_APPLY=$e run)"; echo "B32 fin_apply=$e $(DS_FIN_APPLY=$e run --batch 32)"; done; done > gpurun_out/r06t/ab2.txt 2>&1
sort gpurun_out/r06t/ab2.txt
