#!/bin/bash
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=${DS_LIB:-$R/tumblr_emotions_amd/libds_kernels_tuning.so}
mkdir -p gpurun_out/r06p
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for b in 128 64 32; do for i in 1 2 3 4; do for e in 0 2 3; do echo "B$b batch_bn=$e $(DS_BATCH_BN=$e run --batch $b)"; done; done; done > gpurun_out/r06p/ab2.txt 2>&1
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r06p/ab2.txt"):
    a = l.split()
    d[(a[0], a[1])].append(float(a[2]))
for k in sorted(d): print(k, " ".join("%.3f" % v for v in d[k]), "median %.3f min %.3f" % (statistics.median(d[k]), min(d[k])))
PY
