#!/usr/bin/env python
"""Dispatches of ONE steady-state training step in a rocprofv3 kernel trace: everything between the last two adam_tf_kernel
launches (the whole-trace average also counts the one-time launches of the first step: weight transforms of the frozen
layers, the frozen layers' L2 sums, allocation-time copies).
    python scripts/step_dispatches.py x_results.db"""
import re
import sqlite3
import sys
from collections import Counter


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:60]


cur = sqlite3.connect(sys.argv[1]).cursor()
rows = sorted(cur.execute("select start, end, name from kernels").fetchall())
adam = [i for i, r in enumerate(rows) if "adam_tf" in r[2]]
seg = rows[adam[-2] + 1:adam[-1] + 1]
c = Counter(short(r[2]) for r in seg)
busy = sum(r[1] - r[0] for r in seg) / 1e6
print("# last step: %d dispatches, %.3f ms of kernel time (whole trace: %d dispatches over %d steps = %.1f per step)" % (
    len(seg), busy, len(rows), len(adam), len(rows) / max(len(adam), 1)))
for k, n in c.most_common():
    print("%4d  %s" % (n, k))
