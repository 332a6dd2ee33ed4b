#!/bin/bash
# bash scripts/around.sh NAME [bench flags]: what runs before / after every launch of a kernel (scripts/around.py) in a fresh trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rm -rf /tmp/_ar
rocprofv3 --kernel-trace -d /tmp/_ar -o ar -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-gather --no-conv-timing "$@" > /tmp/_ar.log 2>&1
python $R/scripts/around.py $(ls /tmp/_ar/*.db | head -1) $NAME || tail -5 /tmp/_ar.log
