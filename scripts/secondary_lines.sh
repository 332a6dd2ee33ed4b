R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
out=gpurun_out/${TAG:-r06}_secondary_lines.txt; : > $out
for a in "--mode image --batch 128" "--mode image" "--mode text" "--mode text --batch 64" "--train-all" "--batch 32" "--batch 64" "--batch 128" "--rccl-world1" "--dtype bf16" "--dtype fp8" "--dtype bf16 --batch 128" "--dtype fp8 --batch 128" "--mul3" "--graph --batch 32" ""; do
  echo "ARGS: $a" >> $out
  python bench.py $a --steps 20 --warmup 5 --no-cpu-baseline --no-gather 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(d['value'], d['ms_per_step'], r.get('frac'), d['dtype'])" >> $out
done
cat $out
