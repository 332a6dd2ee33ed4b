#!/bin/bash
# the text tower's forward held back until the image tower has passed a stage (DS_TEXT_GATE = index into TOPOLOGY)
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
mkdir -p gpurun_out/r06x
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do
echo "f32 gate=none $(run)"
for g in 0 1 4 6 8 11; do echo "f32 gate=$g $(DS_TEXT_GATE=$g run)"; done
done > gpurun_out/r06x/gate.txt 2>&1
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r06x/gate.txt"):
    a = l.split()
    if len(a) == 3: d[(a[0], a[1])].append(float(a[2]))
for k in sorted(d): print(k, " ".join("%.3f" % v for v in d[k]), "median %.3f" % statistics.median(d[k]))
PY
