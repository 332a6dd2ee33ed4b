#!/bin/bash
export DS_LIB=${DS_LIB:-$(cd $(dirname $0)/.. && pwd)/tumblr_emotions_amd/libds_kernels_tuning.so}
mkdir -p gpurun_out/r06k
for i in 1 2 3; do for e in 1 0; do
  DS_TEXT_FIRST=$e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 text_first=$e', d['ms_per_step'])"
done; done > gpurun_out/r06k/ab.txt 2>&1
python bench.py --serial-towers --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial towers', d['ms_per_step'])" >> gpurun_out/r06k/ab.txt
python bench.py --mode image --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('image only', d['ms_per_step'])" >> gpurun_out/r06k/ab.txt
for e in 1 0; do
  DS_TEXT_FIRST=$e python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 text_first=$e', d['ms_per_step'])"
done >> gpurun_out/r06k/ab.txt 2>&1
cat gpurun_out/r06k/ab.txt
