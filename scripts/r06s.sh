#!/bin/bash
# full GPU suite + ADVICE r05 #2: the bf16 3x3 dgrads on F(4x4) bf16 pieces (DS_WINO16) at small per-GPU batches
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p gpurun_out/r06s
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06s/suite.txt
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for b in 32 64 128; do for i in 1 2 3; do for e in 1 0; do echo "bf16_B$b wino16=$e $(DS_WINO16=$e run --dtype bf16 --batch $b)"; done; done; done > gpurun_out/r06s/wino16.txt 2>&1
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r06s/wino16.txt"):
    a = l.split()
    d[(a[0], a[1])].append(float(a[2]))
for k in sorted(d): print(k, " ".join("%.3f" % v for v in d[k]), "median %.3f" % statistics.median(d[k]))
PY
cat gpurun_out/r06s/suite.txt
