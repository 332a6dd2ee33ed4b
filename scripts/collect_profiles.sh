#!/bin/bash
# Collect the round's measurement artefacts on a GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profiles.sh r01e
# writes gpurun_out/<tag>_{bench.json,kernel_stats.txt,pmc_traffic.txt,pmc.json}; copy them to profiles/.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
rocprofv3 --kernel-trace -d $R/gpurun_out/_kt -o kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-conv-timing > /dev/null 2>&1
python $R/scripts/rocpd_summary.py $(ls $R/gpurun_out/_kt/*.db | head -1) > $R/gpurun_out/${TAG}_kernel_stats.txt
# the same with the Mixed-block branches on one stream: a kernel's begin..end is then its own duration (this is the
# mode bench.py's roofline timing pass runs in)
rocprofv3 --kernel-trace -d $R/gpurun_out/_ks -o ks -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-conv-timing --no-branch-streams > /dev/null 2>&1
python $R/scripts/rocpd_summary.py $(ls $R/gpurun_out/_ks/*.db | head -1) > $R/gpurun_out/${TAG}_kernel_stats_serial.txt
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes, kernel trace only (MI355X_MICROARCH.md)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/_pf -o pf -- python $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-conv-timing --no-gather --no-branch-streams > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/_pw -o pw -- python $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-conv-timing --no-gather --no-branch-streams > /dev/null 2>&1
python $R/scripts/pmc_summary.py $(ls $R/gpurun_out/_pf/*.db | head -1) $(ls $R/gpurun_out/_pw/*.db | head -1) 3 $R/gpurun_out/${TAG}_pmc.json > $R/gpurun_out/${TAG}_pmc_traffic.txt
# matrix-pipe utilisation per kernel (its own PMC pass)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d $R/gpurun_out/_pm -o pm -- python $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-conv-timing --no-gather --no-branch-streams > /dev/null 2>&1
python $R/scripts/mfma_util.py $(ls $R/gpurun_out/_pm/*.db | head -1) > $R/gpurun_out/${TAG}_mfma_util.txt
# the embedding gather at 2^20 tokens: HBM bytes from the counters (5 launches)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/_gf -o gf -- python $R/scripts/gather_pmc.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/_gw -o gw -- python $R/scripts/gather_pmc.py > /dev/null 2>&1
python $R/scripts/pmc_summary.py $(ls $R/gpurun_out/_gf/*.db | head -1) $(ls $R/gpurun_out/_gw/*.db | head -1) 5 > $R/gpurun_out/${TAG}_gather_pmc.txt
rm -rf $R/gpurun_out/_kt $R/gpurun_out/_ks $R/gpurun_out/_pf $R/gpurun_out/_pw $R/gpurun_out/_pm $R/gpurun_out/_gf $R/gpurun_out/_gw
cat $R/gpurun_out/${TAG}_bench.json
head -8 $R/gpurun_out/${TAG}_kernel_stats.txt
head -8 $R/gpurun_out/${TAG}_pmc_traffic.txt
head -8 $R/gpurun_out/${TAG}_mfma_util.txt
grep -E "gather|TOTAL|kernel" $R/gpurun_out/${TAG}_gather_pmc.txt | head -5
