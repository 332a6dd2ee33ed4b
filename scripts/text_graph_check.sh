#!/bin/bash
# text-only step, eager against hipGraph replay (the launch-bound lines follow the host's load: profiles/r05_notes.md)
uptime
for a in "--mode text --graph" "--mode text --batch 64 --graph" "--mode text" "--mode text --batch 64"; do
  echo -n "$a: "
  python bench.py $a --steps 50 --warmup 10 --no-cpu-baseline --no-conv-timing --no-gather 2>&1 | tail -1 | python -c "
import json, sys
l = sys.stdin.readlines()[-1]
try:
    print(json.loads(l).get('ms_per_step'))
except Exception:
    print(l[:200])"
done
