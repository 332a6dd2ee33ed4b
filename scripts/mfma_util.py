#!/usr/bin/env python
"""MFMA utilisation per kernel from one rocprofv3 pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace ...
    python scripts/mfma_util.py x_results.db > profiles/rNN_mfma_util.txt
GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (it reads 8 x the shader clock x duration), so
    clock    = GRBM_GUI_ACTIVE / 8 / kernel duration
    MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
(the gfx94x derived-counter formula with that correction; cross-checked on conv2c dgrad: 177.6 GFLOP = 43.4 M
wave-level v_mfma_f32_32x32x2_f32 x 64 cycles over 1.5 ms x 2.24 GHz x 1024 SIMDs = 80 %).
Counters are summed over a kernel's launches."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    cnt = defaultdict(lambda: defaultdict(float))
    for name, counter, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        cnt[short(name)][counter] += val
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else "kernel_name"
    dur, calls = defaultdict(float), defaultdict(int)
    for name, s, e in cur.execute("select %s, start, end from kernels" % ncol):
        dur[short(name)] += (e - s) * 1e-9
        calls[short(name)] += 1
    print("%-52s %7s %10s %9s %9s %11s" % ("kernel", "calls", "total_ms", "MfmaUtil", "clock_GHz", "mfma_Gops"))
    for k in sorted(cnt, key=lambda k: -dur.get(k, 0)):
        c = cnt[k]
        gui, busy = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if busy <= 0:
            continue
        util = busy / (gui / 8 * 256 * 4) if gui else float("nan")
        clk = gui / 8 / dur[k] / 1e9 if dur.get(k) else float("nan")
        print("%-52s %7d %10.3f %8.1f%% %9.2f %11.3f" % (k[:52], calls[k], dur[k] * 1e3, 100 * util, clk,
                                                        c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) / 1e9))


if __name__ == "__main__":
    main()
