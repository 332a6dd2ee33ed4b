R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
for mode in kt ks; do
  extra=""; [ $mode = ks ] && extra="--no-branch-streams"
  rm -rf $R/gpurun_out/_$mode
  rocprofv3 --kernel-trace -d $R/gpurun_out/_$mode -o $mode -- python $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-conv-timing $extra > /dev/null 2>&1
  db=$(ls $R/gpurun_out/_$mode/*.db | head -1)
  python $R/scripts/rocpd_summary.py $db > $R/gpurun_out/${TAG:-r06}_bf16_${mode}_stats.txt
  [ $mode = kt ] && python $R/scripts/timeline.py $db > $R/gpurun_out/${TAG:-r06}_bf16_timeline.txt
  rm -rf $R/gpurun_out/_$mode
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/_pf -o pf -- python $R/bench.py --dtype bf16 --steps 3 --warmup 0 --no-cpu-baseline --no-conv-timing --no-gather --no-branch-streams > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/_pw -o pw -- python $R/bench.py --dtype bf16 --steps 3 --warmup 0 --no-cpu-baseline --no-conv-timing --no-gather --no-branch-streams > /dev/null 2>&1
python $R/scripts/pmc_summary.py $(ls $R/gpurun_out/_pf/*.db | head -1) $(ls $R/gpurun_out/_pw/*.db | head -1) 3 > $R/gpurun_out/${TAG:-r06}_bf16_pmc_traffic.txt
rm -rf $R/gpurun_out/_pf $R/gpurun_out/_pw
head -50 $R/gpurun_out/${TAG:-r06}_bf16_timeline.txt
