export PYTHONPATH=$GRAFT_REPO_ROOT
for i in 1 2; do
python scratch/conv_bench.py scratch/libcur.so 0,0 2>&1 | grep cfg
python scratch/conv_bench.py scratch/libnew.so 0,0 2>&1 | grep cfg
done
