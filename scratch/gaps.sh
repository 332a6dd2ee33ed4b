cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/_kg -o kg -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-gather --no-conv-timing $EXTRA > /dev/null 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$R/gpurun_out/_kg/*.db")[0])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
ncol = "name" if "name" in cols else "kernel_name"
rows = sorted(cur.execute("select start, end, %s from kernels" % ncol).fetchall())
# take the last 3 steps: find adam kernels as step delimiters
adam = [i for i, r in enumerate(rows) if "adam_tf" in r[2]]
lo, hi = adam[-4] + 1, adam[-1] + 1
seg = rows[lo:hi]
t0, t1 = seg[0][0], max(r[1] for r in seg)
# union of intervals
cover = 0; cur_s, cur_e = seg[0][0], seg[0][1]
gaps = []
for s, e, n in seg[1:]:
    if s > cur_e:
        cover += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cover += cur_e - cur_s
print("3 steps: wall %.3f ms, covered %.3f ms, idle %.3f ms in %d gaps" % ((t1 - t0) / 1e6, cover / 1e6, (t1 - t0 - cover) / 1e6, len(gaps)))
gaps.sort(reverse=True)
for g, n in gaps[:12]:
    print("  gap %.1f us before %s" % (g / 1e3, n[:70]))
import collections
byk = collections.Counter()
for g, n in gaps: byk[n.split("(")[0][:50]] += g
for n, g in byk.most_common(10): print("  total %.1f us before %s" % (g / 1e3, n))
PY
rm -rf $R/gpurun_out/_kg
