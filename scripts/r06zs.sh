#!/bin/bash
# Mixed_5c's four weight gradients enqueued behind a LATER stage of the backward (DS_WGRAD_DEFER=<end point>).  Measured: a loss at every
# stage; the switch was removed again (r06_notes)
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['final_loss'])"; }
for i in 1 2; do
  echo "f32 none $(run)"
  for g in Mixed_5b Mixed_4e Mixed_4b Mixed_3c Mixed_3b Conv2d_2c_3x3 Conv2d_1a_7x7; do echo "f32 $g $(DS_WGRAD_DEFER=$g run)"; done
  echo "bf16 none $(run --dtype bf16)"
  for g in Mixed_4e Mixed_3c Conv2d_2c_3x3; do echo "bf16 $g $(DS_WGRAD_DEFER=$g run --dtype bf16)"; done
  echo "f32 B128 none $(run --batch 128)"
  for g in Mixed_4e Mixed_3c; do echo "f32 B128 $g $(DS_WGRAD_DEFER=$g run --batch 128)"; done
done | sort
