#!/bin/bash
# bash scripts/step_dispatches.sh TAG [bench flags]  ->  gpurun_out/<TAG>_step_dispatches.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rm -rf /tmp/_sd
rocprofv3 --kernel-trace -d /tmp/_sd -o sd -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-conv-timing "$@" > /tmp/_sd.log 2>&1
mkdir -p $R/gpurun_out
python $R/scripts/step_dispatches.py $(ls /tmp/_sd/*.db | head -1) > $R/gpurun_out/${TAG}_step_dispatches.txt || tail -5 /tmp/_sd.log
head -50 $R/gpurun_out/${TAG}_step_dispatches.txt
