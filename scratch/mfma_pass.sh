cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d $R/gpurun_out/_mu -o mu -- python $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-conv-timing --no-gather --no-branch-streams > /dev/null 2>&1
python $R/scripts/mfma_util.py $(ls $R/gpurun_out/_mu/*.db | head -1) > $R/gpurun_out/${1}_mfma_util.txt
rm -rf $R/gpurun_out/_mu
cat $R/gpurun_out/${1}_mfma_util.txt
