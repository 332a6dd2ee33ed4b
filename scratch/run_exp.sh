for cfg in 0,0 2,2 2,3 2,4 1,2 1,4; do python scratch/conv_bench.py scratch/libcur.so $cfg 2>&1 | grep cfg; done
