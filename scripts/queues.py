#!/usr/bin/env python
"""Which hardware queue ran which kernels?  From a rocprofv3 kernel trace (rocpd sqlite): for the last full step, per
queue / stream id: dispatch count, first start, last end (ms from the step's first kernel) and its first kernels."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:40]


cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("columns:", cols)
ncol = "name" if "name" in cols else "kernel_name"
extra = [c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols]
rows = sorted(cur.execute("select start, end, %s%s from kernels" % (ncol, "".join(", " + c for c in extra))).fetchall())
adam = [i for i, r in enumerate(rows) if "adam_tf" in r[2]]
seg = rows[adam[-2] + 1:adam[-1] + 1]
t0 = seg[0][0]
by = defaultdict(list)
for r in seg:
    by[tuple(r[3:])].append(r)
for k, v in sorted(by.items(), key=lambda kv: kv[1][0][0]):
    print("%s %s: %4d kernels, %7.3f .. %7.3f ms | %s" % (extra, k, len(v), (v[0][0] - t0) / 1e6, (max(x[1] for x in v) - t0) / 1e6,
                                                       ", ".join(short(x[2]) for x in v[:4])))
