#!/bin/bash
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
mkdir -p gpurun_out/r06x
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do
for e in 2 3 0; do echo "B256 batch_bn=$e $(DS_BATCH_BN=$e run)"; done
for m in 1 0; do echo "B64 side=$m $(run --batch 64 --side-mode $m)"; echo "B128 side=$m $(run --batch 128 --side-mode $m)"; done
for r in 1 2 4; do echo "B32 lstm_rows=$r $(run --batch 32 --lstm-rows $r)"; done
done > gpurun_out/r06x/misc2.txt 2>&1
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r06x/misc2.txt"):
    a = l.split()
    if len(a) == 3: d[(a[0], a[1])].append(float(a[2]))
for k in sorted(d): print(k, " ".join("%.3f" % v for v in d[k]), "median %.3f" % statistics.median(d[k]))
PY
