#!/bin/bash
# the text tower's BPTT enqueued from inside the image tower's backward, behind a stage (DS_TEXT_BWD_GATE).  Measured, a loss at every
# stage; the hook (TextTowerFunction.backward leaving a closure the image backward calls) was removed again: r06_notes
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['final_loss'])"; }
for i in 1 2; do
  echo "f32 none $(run)"
  for g in Mixed_5b Mixed_4e Mixed_4b Mixed_3c Mixed_3b Conv2d_2c_3x3; do
    echo "f32 $g $(DS_TEXT_BWD_GATE=$g run)"
    echo "f32 $g rows4 $(DS_TEXT_BWD_GATE=$g run --lstm-rows 4)"
  done
  echo "bf16 none $(run --dtype bf16)"
  for g in Mixed_4e Mixed_3c; do echo "bf16 $g $(DS_TEXT_BWD_GATE=$g run --dtype bf16)"; done
done | sort
