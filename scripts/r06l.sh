#!/bin/bash
mkdir -p gpurun_out/r06l
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2; do
for a in "" "--lstm-rows 1" "--lstm-rows 2" "--lstm-rows 4" "--serial-towers" "--serial-towers --lstm-rows 1" "--serial-towers --lstm-rows 2"; do echo "[$a] $(run $a)"; done
done > gpurun_out/r06l/lstm.txt 2>&1
cat gpurun_out/r06l/lstm.txt
