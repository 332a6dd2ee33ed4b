export PYTHONPATH=$GRAFT_REPO_ROOT
for e in 0 6 0 6; do python scratch/conv_bench.py scratch/libexp$e.so 0,0 2>&1 | grep cfg; done
