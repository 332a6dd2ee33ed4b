#!/bin/bash
# usage: ms.sh <bench args...>  -> prints ms_per_step
python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing --no-live-traffic 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
