import re, sqlite3, sys
db = sys.argv[1]; pat = sys.argv[2]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, queue_id, stream_id from kernels order by start").fetchall()
prev_end = None
out = []
for i, r in enumerate(rows):
    if re.search(pat, r[0]):
        # overlapping kernels at start?
        ov = [re.sub(r"\(anonymous namespace\)::|^void ", "", q[0])[:30] for q in rows[max(0,i-6):i+6] if q is not r and q[1] < r[2] and q[2] > r[1]]
        out.append(((r[2]-r[1])/1e3, r[3]//256 if r[3] else 0, r[6], r[7], ov))
for o in out[-70:]: print("%.1f us grid %d q %s s %s overlaps %s" % o)
