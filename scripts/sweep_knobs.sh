#!/bin/bash
export DS_LIB=${DS_LIB:-$(cd $(dirname $0)/.. && pwd)/tumblr_emotions_amd/libds_kernels_tuning.so}      # the DS_* A/B switches are honoured beside the tuning build only
# A/B of bench flags / environment knobs, interleaved on one box:  bash scripts/sweep_knobs.sh "<flags A>" "<flags B>" [pairs]
run() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
A=${1:-}; B=${2:---no-pool-first}; N=${3:-3}
for i in $(seq $N); do echo "A [$A] $(run $A)"; echo "B [$B] $(run $B)"; done
