"""A/B of the embedding gather at 2^20 tokens (bench.py's `gather` workload): DS_GATHER_OUTORDER = 0 (walk the id list),
1 (walk the time-major OUTPUT rows, non-temporal stores), 2 (the same with plain stores); prints us and HBM-visible TB/s
for the time-major and the batch-major output, uniform and Zipf ids.  Run once per setting (the switch is read once)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tumblr_emotions_amd import _lib, ops  # noqa: E402

if os.environ.get("DS_LIB"):        # A/B runs of kernel variants on one box
    _lib.LIB_PATH = os.environ["DS_LIB"]

V, D, B, T = 10000, 300, 8192, 128
table = torch.randn(V + 1, D, device="cuda")
out = torch.empty(T * B, D, device="cuda")
rng = np.random.RandomState(0)
zipf = np.minimum(rng.zipf(1.2, size=(B, T)) - 1, V).astype(np.int64)
for name, ids in (("uniform", torch.randint(0, V + 1, (B, T), device="cuda", dtype=torch.int64)),
                  ("zipf", torch.from_numpy(zipf).cuda())):
    for tm in (True, False):
        for _ in range(3):
            ops.gather_rows(table, ids, out, B, T, D, time_major=tm)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gather_rows(table, ids, out, B, T, D, time_major=tm)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        ref = table[ids.t().reshape(-1)] if tm else table[ids.reshape(-1)]
        ok = torch.equal(out, ref)
        hbm = B * T * (D * 4 + 8)
        print("OUTORDER=%s %-8s %s: %7.1f us  %.2f TB/s HBM-visible (%.3f of 8)  bit-exact %s"
              % (os.environ.get("DS_GATHER_OUTORDER", "default"), name, "time-major " if tm else "batch-major", us,
                 hbm / us / 1e6, hbm / us / 1e6 / 8, ok))
