#!/bin/bash
# rocprofv3 kernel trace of one bench configuration -> gpurun_out/<tag>_kernel_stats.txt
#   bash scripts/ktrace.sh TAG [bench flags...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rm -rf $R/gpurun_out/_kt
rocprofv3 --kernel-trace -d $R/gpurun_out/_kt -o kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-conv-timing "$@" > /dev/null 2>&1
python $R/scripts/rocpd_summary.py $(ls $R/gpurun_out/_kt/*.db | head -1) > $R/gpurun_out/${TAG}_kernel_stats.txt
rm -rf $R/gpurun_out/_kt
head -${KT_LINES:-26} $R/gpurun_out/${TAG}_kernel_stats.txt | cut -c1-170
