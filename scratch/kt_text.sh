cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/_kx -o kx -- python $R/bench.py --mode text --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing > /dev/null 2>&1
python $R/scripts/rocpd_summary.py $(ls $R/gpurun_out/_kx/*.db | head -1) > $R/gpurun_out/r02j_kernel_stats_text_b64.txt
rm -rf $R/gpurun_out/_kx
head -30 $R/gpurun_out/r02j_kernel_stats_text_b64.txt | cut -c1-70,115-160
