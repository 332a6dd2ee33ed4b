export PYTHONPATH=$GRAFT_REPO_ROOT
export DS_CONV_PATH=l
for e in 0 9 0 9; do python scratch/conv_bench.py scratch/libexp$e.so 0,0 2>&1 | grep cfg; done
