# interleaved A/B: $1 = flag for variant B, $2 = repetitions
for i in $(seq 1 ${2:-2}); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A', d['ms_per_step'])"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing $1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B $1', d['ms_per_step'])"
done
