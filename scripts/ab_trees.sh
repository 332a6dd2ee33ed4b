#!/bin/bash
# Interleaved in-box A/B of whole trees / flag sets (two boxes differ by +-3 %, so only in-box comparisons count):
#   bash scripts/ab_trees.sh REPS "dir|flags" "dir|flags" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
REPS=$1; shift
for i in $(seq 1 $REPS); do
  for spec in "$@"; do
    d=${spec%%|*}; f=${spec#*|}
    ms=$(cd $R/$d && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing $f 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$d [$f] $ms"
  done
done
