#!/bin/bash
# NOTE: needs an ablation build of conv_igemm.hip (-DDS_BF16D_NOSTORE: the epilogue's output stores compiled out) that was
# removed again with the experiment; kept as the record of how profiles/r06_notes.md's ablation numbers were taken.
# ablation: the register-direct bf16 kernels WITHOUT their output stores (results wrong, timing only): what do the stores cost?
R=$(cd $(dirname $0)/.. && pwd)
T=$R/tumblr_emotions_amd/libds_kernels_tuning.so
A=$R/tumblr_emotions_amd/csrc/build_tuning/libds_tuning_nostore.so
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2; do echo "bf16 stores $(DS_LIB=$T run --dtype bf16)"; echo "bf16 no-stores $(DS_LIB=$A run --dtype bf16)"; done
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
for L in $T $A; do rm -rf /tmp/_kt; DS_LIB=$L rocprofv3 --kernel-trace -d /tmp/_kt -o kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-conv-timing --no-branch-streams --dtype bf16 > /dev/null 2>&1; echo "== $L"; python $R/scripts/rocpd_summary.py $(ls /tmp/_kt/*.db | head -1) | grep "conv_bf16d" | cut -c1-50,100-170 | head -12; done
