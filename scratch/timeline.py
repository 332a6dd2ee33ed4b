"""Dump the dispatch timeline of the last training step in a rocpd kernel trace."""
import re, sqlite3, sys
db = sys.argv[1]; nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 120
con = sqlite3.connect(db); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
scol = "stream_id" if "stream_id" in cols else None
sel = "name, start, end, grid_x, grid_y, grid_z" + ((", " + qcol) if qcol else "") + ((", " + scol) if scol else "")
rows = cur.execute("select %s from kernels order by start" % sel).fetchall()
# find the start of the last step: the last dispatch of the stem kernel (fold variant: grid large, first conv)
idx = [i for i, r in enumerate(rows) if "gather" in r[0]]
start = idx[-1] - 3 if idx else 0
t0 = rows[start][1]
for r in rows[start:start + nshow]:
    n = re.sub(r"\(anonymous namespace\)::|^void ", "", r[0])[:60]
    print("%9.1f %9.1f %8.1f  g=%-5d,%-3d,%-2d q=%s %s" % ((r[1]-t0)/1e3, (r[2]-t0)/1e3, (r[2]-r[1])/1e3, r[3]//256 if r[3] else 0, r[4], r[5], r[6:] , n))
