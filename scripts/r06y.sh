#!/bin/bash
# rows per pass of the fixed-column streaming kernels (-DDS_BN_ROWS=2 shipped / 3 / 4 variants), step level
R=$(cd $(dirname $0)/.. && pwd)
T=$R/tumblr_emotions_amd/libds_kernels_tuning.so
mkdir -p gpurun_out/r06y
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
DS_LIB=$R/tumblr_emotions_amd/csrc/build_tuning/libds_tuning_rows4.so timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "batch_norm" 2>&1 | tail -2
for i in 1 2 3; do for r in 2 3 4; do
  L=$T; [ $r != 2 ] && L=$R/tumblr_emotions_amd/csrc/build_tuning/libds_tuning_rows$r.so
  echo "f32 rows=$r $(DS_LIB=$L run)"; echo "bf16 rows=$r $(DS_LIB=$L run --dtype bf16)"; echo "B64 rows=$r $(DS_LIB=$L run --batch 64)"
done; done > gpurun_out/r06y/rows.txt 2>&1
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r06y/rows.txt"):
    a = l.split()
    if len(a) == 3: d[(a[0], a[1])].append(float(a[2]))
for k in sorted(d): print(k, " ".join("%.3f" % v for v in d[k]), "median %.3f" % statistics.median(d[k]))
PY
