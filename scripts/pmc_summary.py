#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes).  FETCH_SIZE on gfx950 counts 128-B requests at 64 B for wide coalesced
reads, so the read side is reported raw AND doubled (the guide's correction); WRITE_SIZE calibrates 1:1
(checked here on torch's 14.72 MB fill: 14377.5 KB reported).

    python scripts/pmc_summary.py fetch.db write.db STEPS > profiles/rNN_pmc_traffic.txt
also writes profiles/<prefix>.json when a 4th argument (output json path) is given.
"""
import json
import re
import sqlite3
import sys
from collections import defaultdict

# the launches bench.py's roofline pass brackets (ops.ConvTimer: everything that goes through ds_conv_igemm / ds_conv_wino /
# ds_conv_wino4 / ds_conv_stem / ds_conv_bf16 / ds_conv_fp8)
CONV_FAMILY = ("conv_igemm_kernel", "conv_glds_kernel", "gemm_wide_kernel", "conv_wino_kernel", "conv_wino4_kernel",
               "conv_stem_kernel", "conv_bf16_kernel", "conv_bf16d_kernel", "conv_fp8d_kernel")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return name


def load(db, counter):
    cur = sqlite3.connect(db).cursor()
    agg = defaultdict(lambda: [0, 0.0])
    for name, val in cur.execute("select kernel_name, value from counters_collection where counter_name=?", (counter,)):
        a = agg[short(name)]
        a[0] += 1
        a[1] += val * 1024.0          # KB -> bytes
    return agg


def main():
    fdb, wdb, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    f, w = load(fdb, "FETCH_SIZE"), load(wdb, "WRITE_SIZE")
    names = sorted(set(f) | set(w), key=lambda n: -(2 * f.get(n, [0, 0])[1] + w.get(n, [0, 0])[1]))
    print("# per training step (%d profiled steps); bytes in MB; FETCHx2 = gfx950 correction" % steps)
    print("%-60s %8s %12s %12s %12s %14s" % ("kernel", "calls", "FETCH_raw", "FETCHx2", "WRITE", "HBM(2F+W)"))
    tot = [0.0, 0.0]
    out = {}
    for n in names:
        c = max(f.get(n, [0, 0])[0], w.get(n, [0, 0])[0]) / steps
        fb, wb = f.get(n, [0, 0])[1] / steps, w.get(n, [0, 0])[1] / steps
        tot[0] += fb
        tot[1] += wb
        out[n] = dict(calls_per_step=c, fetch_raw_bytes=fb, fetch_corrected_bytes=2 * fb, write_bytes=wb)
        print("%-60s %8.1f %12.1f %12.1f %12.1f %14.1f" % (n[:60], c, fb / 1e6, 2 * fb / 1e6, wb / 1e6, (2 * fb + wb) / 1e6))
    print("%-60s %8s %12.1f %12.1f %12.1f %14.1f" % ("TOTAL", "", tot[0] / 1e6, 2 * tot[0] / 1e6, tot[1] / 1e6, (2 * tot[0] + tot[1]) / 1e6))
    if len(sys.argv) > 4:
        conv = [v for k, v in out.items() if k.startswith(CONV_FAMILY)]
        calls = sum(v["calls_per_step"] for v in conv)
        hbm = sum(v["fetch_corrected_bytes"] + v["write_bytes"] for v in conv)
        json.dump(dict(kernel=" + ".join(CONV_FAMILY), launches_per_step=calls, hbm_bytes_per_step=hbm,
                       hbm_bytes_per_launch=hbm / max(calls, 1), per_kernel=out,
                       note="FETCH_SIZE doubled per MI355X_MICROARCH.md; WRITE_SIZE as reported"),
                  open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
