cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for i in 1 0; do
DS_WGRAD_DIRECT=$i rocprofv3 --kernel-trace -d $R/gpurun_out/_kt$i -o kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-conv-timing > /dev/null 2>&1
python $R/scripts/rocpd_summary.py $(ls $R/gpurun_out/_kt$i/*.db | head -1) | grep -E "wgrad|splitk|kernels:"
rm -rf $R/gpurun_out/_kt$i
done
