#!/bin/bash
# BatchNorm-backward sums from the Branch_3 pool gradient (ds_maxpool3_bwd_sums) in the 16-bit configurations
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=${DS_LIB:-$R/tumblr_emotions_amd/libds_kernels_tuning.so}
mkdir -p gpurun_out/r06q
python -m pytest tests/test_kernels_gpu.py -x -q -k "emits_the_batch_norm_sums or max_pool" 2>&1 | tail -3 > gpurun_out/r06q/t1.txt
python -m pytest tests/test_model_gpu.py -q -s -k "branch3_pool_gradient" 2>&1 | grep -v "^$" | tail -5 > gpurun_out/r06q/t2.txt
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3 4; do for e in 1 0; do echo "bf16 pool_sums=$e $(DS_POOL_SUMS=$e run --dtype bf16)"; done; done > gpurun_out/r06q/ab.txt 2>&1
for i in 1 2 3; do for e in 1 0; do echo "bf16_B128 pool_sums=$e $(DS_POOL_SUMS=$e run --dtype bf16 --batch 128)"; done; done >> gpurun_out/r06q/ab.txt 2>&1
for i in 1 2; do for e in 1 0; do echo "fp8 pool_sums=$e $(DS_POOL_SUMS=$e run --dtype fp8)"; done; done >> gpurun_out/r06q/ab.txt 2>&1
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r06q/ab.txt"):
    a = l.split()
    d[(a[0], a[1])].append(float(a[2]))
for k in sorted(d): print(k, " ".join("%.3f" % v for v in d[k]), "median %.3f" % statistics.median(d[k]))
PY
unset DS_LIB
KT_LINES=60 bash scripts/ktrace.sh r06q_bf16 --dtype bf16 > /dev/null 2>&1; grep -i "maxpool3s1\|kernels:" gpurun_out/r06q_bf16_kernel_stats.txt | cut -c1-160
cat gpurun_out/r06q/t*.txt
