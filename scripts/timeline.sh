#!/bin/bash
# bash scripts/timeline.sh TAG [bench flags]  ->  gpurun_out/<TAG>_timeline.txt  (rocprofv3 kernel trace -> scripts/timeline.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rm -rf /tmp/_tl
rocprofv3 --kernel-trace -d /tmp/_tl -o tl -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-gather --no-conv-timing "$@" > /tmp/_tl.log 2>&1
mkdir -p $R/gpurun_out
python $R/scripts/timeline.py $(ls /tmp/_tl/*.db | head -1) > $R/gpurun_out/${TAG}_timeline.txt 2>&1 || tail -5 /tmp/_tl.log
head -${TL_LINES:-45} $R/gpurun_out/${TAG}_timeline.txt
