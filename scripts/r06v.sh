#!/bin/bash
# split-K F(4x4) (ds_conv_wino4_splitk): tests, per-layer times, step A/B at small batches
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=${DS_LIB:-$R/tumblr_emotions_amd/libds_kernels_tuning.so}
mkdir -p gpurun_out/r06v
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "split_over_workgroups or winograd" 2>&1 | tail -5 > gpurun_out/r06v/t1.txt
cat gpurun_out/r06v/t1.txt
if grep -q "failed\|error" gpurun_out/r06v/t1.txt; then exit 1; fi
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -s -k "split_k or b32 or oracle" 2>&1 | grep -v "^$" | tail -6 > gpurun_out/r06v/t2.txt
cat gpurun_out/r06v/t2.txt
if grep -q "failed\|error" gpurun_out/r06v/t2.txt; then exit 1; fi
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for b in 32 64 128; do for i in 1 2 3; do for e in 1 0; do echo "B$b splitk=$e $(DS_SPLITK=$e run --batch $b)"; done; done; done > gpurun_out/r06v/ab.txt 2>&1
for i in 1 2; do for e in 1 0; do echo "B256 splitk=$e $(DS_SPLITK=$e run)"; done; done >> gpurun_out/r06v/ab.txt 2>&1
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r06v/ab.txt"):
    a = l.split()
    if len(a) == 3: d[(a[0], a[1])].append(float(a[2]))
for k in sorted(d): print(k, " ".join("%.3f" % v for v in d[k]), "median %.3f" % statistics.median(d[k]))
PY
