#!/bin/bash
# usage: pmc_one.sh "<counters>" <kernel-name substring> -- <command...>   -> mean counter values per dispatch of that kernel
C="$1"; KN="$2"; shift 3
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d /tmp/pmc_XXXX)
rocprofv3 --pmc $C --kernel-trace -d $D -o p -- "$@" > $D/log.txt 2>&1
python3 - "$D" "$KN" <<'PY'
import sys, glob, sqlite3, collections
dbs = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)
if not dbs:
    print("no database:", open(sys.argv[1] + "/log.txt").read()[-600:])
    sys.exit(0)
cur = sqlite3.connect(dbs[0]).cursor()
acc = collections.defaultdict(lambda: [0, 0.0])
for name, cname, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
    if sys.argv[2] in name:
        a = acc[(name.replace("(anonymous namespace)::", "")[:60], cname)]
        a[0] += 1
        a[1] += val
for (k, c), (n, v) in sorted(acc.items()):
    print("%-60s %-24s n=%d  mean=%.4g" % (k, c, n, v / n))
PY
rm -rf $D
