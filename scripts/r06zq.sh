#!/bin/bash
# B = 32: is the eager step host-bound?  ms/step beside the host's time to enqueue a step
run() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'host', d['config'].get('host_enqueue_ms_per_step'))"; }
for i in 1 2; do
  for b in 16 32 64 128; do
    echo "B=$b eager $(run --batch $b)"
    echo "B=$b graph $(run --batch $b --graph)"
  done
  echo "B=32 eager side1 $(run --batch 32 --side-mode 1)"
  echo "B=32 eager side0 $(run --batch 32 --side-mode 0)"
  echo "B=32 eager no-branch-streams $(run --batch 32 --no-branch-streams)"
  echo "B=32 eager serial-towers $(run --batch 32 --serial-towers)"
done | sort
