#!/bin/bash
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
mkdir -p gpurun_out/r06x
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do
echo "f32 default $(run)"
echo "f32 lstm_shared_cu $(DS_LSTM_SHARED_CU=1 run)"
echo "f32 wide_cost1 $(DS_WIDE_COST=1 run)"
echo "f32 wide_cost2 $(DS_WIDE_COST=2 run)"
echo "f32 wide_minwgs64 $(DS_WIDE_MINWGS=64 run)"
echo "f32 wide_minwgs256 $(DS_WIDE_MINWGS=256 run)"
echo "f32 wgrad_occ2 $(DS_WGRAD_OCC=2 run)"
done > gpurun_out/r06x/misc.txt 2>&1
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r06x/misc.txt"):
    a = l.split()
    if len(a) == 3: d[(a[0], a[1])].append(float(a[2]))
for k in sorted(d): print(k, " ".join("%.3f" % v for v in d[k]), "median %.3f" % statistics.median(d[k]))
PY
