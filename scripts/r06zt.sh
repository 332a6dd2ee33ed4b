#!/bin/bash
# CU-masked side streams (hipExtStreamCreateWithCUMask): confining the side work to a share of the CUs.  Measured: the step DOUBLES with any
# masked stream in the process; the hook in streams.py was removed again (r06_notes)
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
run() { timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['final_loss'])"; }
for i in 1 2; do
  echo "f32 none $(run)"
  for n in 32 64 128; do echo "f32 wgrad-stream=$n $(DS_CUMASK_side0=$n run)"; done
  for n in 64 96 128 192; do echo "f32 branch-side-stream=$n $(DS_CUMASK_side1=$n run)"; done
  for n in 32 64; do echo "f32 text=$n $(DS_CUMASK_text=$n run)"; done
  echo "f32 side1=128+side0=64 $(DS_CUMASK_side1=128 DS_CUMASK_side0=64 run)"
done | sort
