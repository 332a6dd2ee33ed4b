#!/usr/bin/env python
"""What runs right before / after every launch of a named kernel in the last step of a rocprofv3 trace (same queue)?
    python scripts/around.py x_results.db copyBuffer"""
import re, sqlite3, sys
from collections import Counter
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:44]
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = sorted(cur.execute("select start, end, name, queue_id, grid_x, workgroup_x from kernels").fetchall())
adam = [i for i, r in enumerate(rows) if "adam_tf" in r[2]]
seg = rows[adam[-2] + 1:adam[-1] + 1]
byq = {}
for r in seg:
    byq.setdefault(r[3], []).append(r)
c = Counter()
for q, v in byq.items():
    for i, r in enumerate(v):
        if sys.argv[2] in r[2]:
            prev = short(v[i - 1][2]) if i else "-"
            nxt = short(v[i + 1][2]) if i + 1 < len(v) else "-"
            c[(q, prev, nxt, r[4])] += 1
for k, n in c.most_common(40):
    print(n, k)
