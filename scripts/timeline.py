#!/usr/bin/env python
"""Where does a step's wall time go?  From a rocprofv3 kernel trace (rocpd sqlite): take the last full step (between two
adam_tf_kernel launches), cut it into intervals at every kernel start / end, and attribute each interval to the set of
kernels running in it.  Prints: time with 0 / 1 / 2 / 3+ kernels in flight, and the kernels that run ALONE (nothing
else in flight) ranked by the wall time they hold -- those are the serial part of the step.
    python scripts/timeline.py x_results.db [steps_back]"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:60]


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else "kernel_name"
    rows = sorted(cur.execute("select start, end, %s from kernels" % ncol).fetchall())
    adam = [i for i, r in enumerate(rows) if "adam_tf" in r[2]]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lo, hi = adam[-1 - back] + 1, adam[-back] + 1
    seg = rows[lo:hi]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    ev = []
    for i, (s, e, n) in enumerate(seg):
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    live = set()
    by_count = defaultdict(float)
    alone = defaultdict(float)
    pairs = defaultdict(float)
    prev = t0
    for t, kind, i in ev:
        if t > prev:
            dt = t - prev
            by_count[min(len(live), 3)] += dt
            if len(live) == 1:
                alone[short(seg[next(iter(live))][2])] += dt
            elif len(live) >= 2:
                pairs[" + ".join(sorted(short(seg[j][2])[:28] for j in live)[:3])] += dt
            prev = t
        if kind:
            live.add(i)
        else:
            live.discard(i)
    wall = (t1 - t0) / 1e6
    print("step: %d kernels, wall %.3f ms" % (len(seg), wall))
    marks = ("gather_rows", "lstm_seq", "avgpool_dropout", "softmax_ce", "conv_stem", "adam_tf", "wgrad_direct")
    print("landmarks (ms from the step's first kernel):")
    for s_, e_, n_ in seg:
        if any(m in n_ for m in marks):
            print("  %7.3f .. %7.3f  %s" % ((s_ - t0) / 1e6, (e_ - t0) / 1e6, short(n_)))
    for k in sorted(by_count):
        print("  %s kernels in flight: %7.3f ms (%4.1f %%)" % ("3+" if k == 3 else k, by_count[k] / 1e6, 100 * by_count[k] / 1e6 / wall))
    print("kernels running ALONE, by wall time held:")
    for n, t in sorted(alone.items(), key=lambda kv: -kv[1])[:22]:
        print("  %7.3f ms  %s" % (t / 1e6, n))
    print("most common concurrent sets:")
    for n, t in sorted(pairs.items(), key=lambda kv: -kv[1])[:12]:
        print("  %7.3f ms  %s" % (t / 1e6, n))


if __name__ == "__main__":
    main()
