#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace: per-kernel calls / total / average,
i.e. the `--stats` table, plus per-shape detail for the conv kernel.  Usage:
    python scripts/rocpd_summary.py gpurun_out/prof/x_results.db [--skip-first N] > profiles/rNN_kernel_stats.txt
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) < 110 else name[:107] + "..."


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, start, end, grid_x, grid_y, workgroup_x from kernels order by start" % name_col).fetchall()
    agg = defaultdict(lambda: [0, 0])
    for (n, s, e, *_rest) in rows:
        a = agg[short(n)]
        a[0] += 1
        a[1] += e - s
    total = sum(v[1] for v in agg.values())
    span = rows[-1][2] - rows[0][1] if rows else 0
    print("# %s" % db)
    print("# kernels: %d dispatches, %.3f ms busy, %.3f ms first-start..last-end" % (len(rows), total / 1e6, span / 1e6))
    print("%-112s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-112s %8d %12.3f %10.2f %6.2f%%" % (n, c, t / 1e6, t / 1e3 / c, 100.0 * t / max(total, 1)))


if __name__ == "__main__":
    main()
