#!/bin/bash
# do the engine's older switches still pay with the round's kernels?  (each OFF against the default)
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
mkdir -p gpurun_out/r06x
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2; do
echo "f32 default $(run)"
echo "f32 no_pool_first $(run --no-pool-first)"
echo "f32 no_bwd_sums $(run --no-bwd-sums)"
echo "f32 bnb0 $(DS_BNB=0 run)"
echo "f32 stem_sums0 $(DS_STEM_SUMS=0 run)"
echo "f32 zcat0 $(DS_ZCAT=0 run)"
echo "f32 no_wino4 $(run --no-wino4)"
echo "f32 fuse_b3_off $(run --no-fuse-b3)"
echo "f32 stem_pool_off $(run --no-stem-pool)"
echo "f32 lstm_sort0 $(DS_LSTM_SORT=0 run)"
echo "f32 serial_towers $(run --serial-towers)"
done > gpurun_out/r06x/switches.txt 2>&1
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r06x/switches.txt"):
    a = l.split()
    if len(a) == 3: d[(a[0], a[1])].append(float(a[2]))
for k in sorted(d, key=lambda k: statistics.median(d[k])): print(k, " ".join("%.3f" % v for v in d[k]), "median %.3f" % statistics.median(d[k]))
PY
