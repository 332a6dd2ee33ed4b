for plan in "side1,text,comm,side0" "text,side1,comm,side0" "side1,x,text,comm,side0" "x,side1,text,comm,side0" "side1,x,x,text,comm,side0" "x,x,side1,text,comm,side0" "x,side1,x,text,comm,side0" "side1,comm,text,side0" "x,x,x,side1,text,comm,side0" "side1,x,x,x,text,comm,side0"; do
export DS_LIB=${DS_LIB:-$(cd $(dirname $0)/.. && pwd)/tumblr_emotions_amd/libds_kernels_tuning.so}      # the DS_* A/B switches are honoured beside the tuning build only
  export DS_STREAM_PLAN=$plan
  ms=$(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$plan $ms"
done
