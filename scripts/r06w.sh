#!/bin/bash
# the block-input gradient in two tensors (split_dout): tests, then A/B on the 16-bit labels
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=${DS_LIB:-$R/tumblr_emotions_amd/libds_kernels_tuning.so}
mkdir -p gpurun_out/r06w
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "two_tensors or batch_norm or max_pool" 2>&1 | tail -5 > gpurun_out/r06w/t1.txt
cat gpurun_out/r06w/t1.txt
if grep -q "failed\|error" gpurun_out/r06w/t1.txt; then exit 1; fi
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -s -k "two_tensors or pool_gradient or 16_bit or bf16 or fp8" 2>&1 | grep -v "^$" | tail -12 > gpurun_out/r06w/t2.txt
cat gpurun_out/r06w/t2.txt
if grep -q "failed\|error" gpurun_out/r06w/t2.txt; then exit 1; fi
timeout 600 python -m pytest tests/test_golden_gpu.py -x -q 2>&1 | tail -3
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do for e in 1 0; do echo "bf16 split_dout=$e $(DS_SPLIT_DOUT=$e run --dtype bf16)"; echo "bf16_B128 split_dout=$e $(DS_SPLIT_DOUT=$e run --dtype bf16 --batch 128)"; done; done > gpurun_out/r06w/ab.txt 2>&1
for i in 1 2; do for e in 1 0; do echo "fp8 split_dout=$e $(DS_SPLIT_DOUT=$e run --dtype fp8)"; echo "fp8_B128 split_dout=$e $(DS_SPLIT_DOUT=$e run --dtype fp8 --batch 128)"; done; done >> gpurun_out/r06w/ab.txt 2>&1
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r06w/ab.txt"):
    a = l.split()
    if len(a) == 3: d[(a[0], a[1])].append(float(a[2]))
for k in sorted(d): print(k, " ".join("%.3f" % v for v in d[k]), "median %.3f" % statistics.median(d[k]))
PY
