cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/_kb -o kb -- python $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-conv-timing --no-branch-streams > /dev/null 2>&1
python $R/scripts/rocpd_summary.py $(ls $R/gpurun_out/_kb/*.db | head -1) > $R/gpurun_out/r02h_kernel_stats_bf16_serial.txt
rm -rf $R/gpurun_out/_kb
head -32 $R/gpurun_out/r02h_kernel_stats_bf16_serial.txt | cut -c1-70,115-160
