#!/bin/bash
# strided row groups of the skip_masked LSTM launches: A/B against the contiguous assignment (DS_LSTM_CONTIG=1, tuning library)
export DS_LIB=${DS_LIB:-$(cd $(dirname $0)/.. && pwd)/tumblr_emotions_amd/libds_kernels_tuning.so}
mkdir -p gpurun_out/r06n
python -m pytest tests/test_kernels_gpu.py -x -q -k "lstm" 2>&1 | tail -6 > gpurun_out/r06n/t1.txt
python -m pytest tests/test_model_gpu.py -x -q -k "length_sorted" 2>&1 | tail -4 > gpurun_out/r06n/t2.txt
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do echo "f32 strided $(run)"; echo "f32 contig $(DS_LSTM_CONTIG=1 run)"; done > gpurun_out/r06n/ab.txt 2>&1
for r in 2 4; do echo "f32 rows=$r strided $(run --lstm-rows $r)"; done >> gpurun_out/r06n/ab.txt 2>&1
echo "bf16 strided $(run --dtype bf16)" >> gpurun_out/r06n/ab.txt; echo "bf16 contig $(DS_LSTM_CONTIG=1 run --dtype bf16)" >> gpurun_out/r06n/ab.txt
echo "B128 strided $(run --batch 128)" >> gpurun_out/r06n/ab.txt; echo "B128 contig $(DS_LSTM_CONTIG=1 run --batch 128)" >> gpurun_out/r06n/ab.txt
echo "text strided $(run --mode text --lstm-rows 4)" >> gpurun_out/r06n/ab.txt; echo "text contig $(DS_LSTM_CONTIG=1 run --mode text --lstm-rows 4)" >> gpurun_out/r06n/ab.txt
cat gpurun_out/r06n/*.txt
