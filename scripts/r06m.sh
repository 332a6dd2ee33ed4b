#!/bin/bash
export DS_LIB=${DS_LIB:-$(cd $(dirname $0)/.. && pwd)/tumblr_emotions_amd/libds_kernels_tuning.so}
mkdir -p gpurun_out/r06m
python -m pytest tests/test_kernels_gpu.py -x -q -k "lstm" 2>&1 | tail -12 > gpurun_out/r06m/t1.txt
python -m pytest tests/test_model_gpu.py tests/test_frontends_gpu.py -x -q -k "length_sorted or text or joint or concat or front" 2>&1 | tail -8 > gpurun_out/r06m/t2.txt
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do for e in 1 0; do echo "f32 sort=$e $(DS_LSTM_SORT=$e run)"; done; done > gpurun_out/r06m/ab.txt 2>&1
for e in 1 0; do echo "bf16 sort=$e $(DS_LSTM_SORT=$e run --dtype bf16)"; echo "text B256 sort=$e $(DS_LSTM_SORT=$e run --mode text)"; echo "text B64 sort=$e $(DS_LSTM_SORT=$e run --mode text --batch 64)"; echo "B32 sort=$e $(DS_LSTM_SORT=$e run --batch 32)"; done >> gpurun_out/r06m/ab.txt 2>&1
cat gpurun_out/r06m/*.txt
