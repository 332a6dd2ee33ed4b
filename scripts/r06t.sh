#!/bin/bash
# BatchNorm finalize + apply as one launch (ds_bn_finalize_apply_relu, ds_bn_bwd_finalize_apply): tests, then A/B
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=${DS_LIB:-$R/tumblr_emotions_amd/libds_kernels_tuning.so}
mkdir -p gpurun_out/r06t
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "as_one_launch or batch_norm" 2>&1 | tail -5 > gpurun_out/r06t/t1.txt
if ! grep -q "passed" gpurun_out/r06t/t1.txt || grep -q "failed" gpurun_out/r06t/t1.txt; then cat gpurun_out/r06t/t1.txt; exit 1; fi
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "as_one_launch or batched_batch_norm or zcat" 2>&1 | tail -5 > gpurun_out/r06t/t2.txt
cat gpurun_out/r06t/t2.txt
if grep -q "failed\|error" gpurun_out/r06t/t2.txt; then exit 1; fi
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do for e in 1 0; do echo "B256 fin_apply=$e $(DS_FIN_APPLY=$e run)"; done; done > gpurun_out/r06t/ab.txt 2>&1
for b in 128 64 32; do for i in 1 2 3; do for e in 1 0; do echo "B$b fin_apply=$e $(DS_FIN_APPLY=$e run --batch $b)"; done; done; done >> gpurun_out/r06t/ab.txt 2>&1
for i in 1 2 3; do for e in 1 0; do echo "bf16 fin_apply=$e $(DS_FIN_APPLY=$e run --dtype bf16)"; done; done >> gpurun_out/r06t/ab.txt 2>&1
for i in 1 2; do for e in 1 0; do echo "bf16_B128 fin_apply=$e $(DS_FIN_APPLY=$e run --dtype bf16 --batch 128)"; done; done >> gpurun_out/r06t/ab.txt 2>&1
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r06t/ab.txt"):
    a = l.split()
    if len(a) == 3: d[(a[0], a[1])].append(float(a[2]))
for k in sorted(d): print(k, " ".join("%.3f" % v for v in d[k]), "median %.3f" % statistics.median(d[k]))
PY
cat gpurun_out/r06t/t1.txt
