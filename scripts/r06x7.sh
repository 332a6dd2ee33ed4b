#!/bin/bash
# the 16-bit label's switches, each OFF against the default (final code, B = 256)
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
mkdir -p gpurun_out/r06x
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing --dtype bf16 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2; do
echo "bf16 default $(run)"
echo "bf16 no_act16 $(run --no-act16)"
echo "bf16 dz16_off $(DS_DZ16=0 run)"
echo "bf16 wino16_off $(DS_WINO16=0 run)"
echo "bf16 pool_sums_off $(DS_POOL_SUMS=0 run)"
echo "bf16 no_bwd_sums $(run --no-bwd-sums)"
echo "bf16 stem_pool_off $(run --no-stem-pool)"
echo "bf16 lstm_sort0 $(DS_LSTM_SORT=0 run)"
echo "bf16 bf16_staged $(run --bf16-staged)"
echo "bf16 pool_first_16 $(DS_POOL_FIRST_16=1 run)"
done > gpurun_out/r06x/switches16.txt 2>&1
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r06x/switches16.txt"):
    a = l.split()
    if len(a) == 3: d[a[1]].append(float(a[2]))
base = statistics.median(d["default"])
for k in sorted(d, key=lambda k: statistics.median(d[k])):
    m = statistics.median(d[k]); print("%-16s %s  median %.3f  (%+.3f)" % (k, " ".join("%.3f" % v for v in d[k]), m, m - base))
PY
