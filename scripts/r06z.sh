#!/bin/bash
# rows in flight per thread of the 3x3/1 max pools on 16-bit input (-DDS_POOL_PF16 = 2 (round 5), 4 (shipped), 6)
R=$(cd $(dirname $0)/.. && pwd)
T=$R/tumblr_emotions_amd/libds_kernels_tuning.so
mkdir -p gpurun_out/r06z
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_fuzz_gpu.py -x -q -k "pool" 2>&1 | tail -2
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do for d in 4 2 6; do
  L=$T; [ $d != 4 ] && L=$R/tumblr_emotions_amd/csrc/build_tuning/libds_tuning_pf$d.so
  echo "bf16 pf=$d $(DS_LIB=$L run --dtype bf16)"; echo "bf16_B128 pf=$d $(DS_LIB=$L run --dtype bf16 --batch 128)"; echo "fp8 pf=$d $(DS_LIB=$L run --dtype fp8)"
done; done > gpurun_out/r06z/pf.txt 2>&1
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r06z/pf.txt"):
    a = l.split()
    if len(a) == 3: d[(a[0], a[1])].append(float(a[2]))
for k in sorted(d): print(k, " ".join("%.3f" % v for v in d[k]), "median %.3f" % statistics.median(d[k]))
PY
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R; rm -rf /tmp/_kt; rocprofv3 --kernel-trace -d /tmp/_kt -o kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-conv-timing --no-branch-streams --dtype bf16 > /dev/null 2>&1; python $R/scripts/rocpd_summary.py $(ls /tmp/_kt/*.db | head -1) | grep "maxpool3_fwd" | cut -c1-60,100-170
