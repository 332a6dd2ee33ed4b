cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/pmc_conv1 -o c1 -- python $R/scratch/conv_bench.py $R/scratch/libcur.so 0,0 2>&1 | tail -15
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM --kernel-trace -d $R/gpurun_out/pmc_conv2 -o c2 -- python $R/scratch/conv_bench.py $R/scratch/libcur.so 0,0 2>&1 | tail -5
