#!/bin/bash
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=${DS_LIB:-$R/tumblr_emotions_amd/libds_kernels_tuning.so}
mkdir -p gpurun_out/r06v
run() { timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for b in 32 16; do for i in 1 2 3 4 5; do for e in 1 0; do echo "B$b splitk=$e $(DS_SPLITK=$e run --batch $b)"; done; done; done > gpurun_out/r06v/ab3.txt 2>&1
for i in 1 2 3; do for e in 1 0; do echo "B32_graph splitk=$e $(DS_SPLITK=$e run --batch 32 --graph)"; echo "B32_image splitk=$e $(DS_SPLITK=$e run --batch 32 --mode image)"; done; done >> gpurun_out/r06v/ab3.txt 2>&1
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r06v/ab3.txt"):
    a = l.split()
    if len(a) == 3: d[(a[0], a[1])].append(float(a[2]))
for k in sorted(d): print(k, " ".join("%.3f" % v for v in d[k]), "median %.3f min %.3f" % (statistics.median(d[k]), min(d[k])))
PY
