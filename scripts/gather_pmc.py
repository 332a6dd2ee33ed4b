#!/usr/bin/env python
"""The embedding gather at 2^20 tokens (north star: "rocprof HBM GB/s on the embedding gather") as a stand-alone
workload for rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o f -- python scripts/gather_pmc.py
The same launch bench.py times (`gather`): D = 300 fp32 rows of a 10 001-row table, time-major output."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402
from tumblr_emotions_amd import ops                     # noqa: E402

V, D, B, T = 10000, 300, 8192, 128
table = torch.randn(V + 1, D, device="cuda")
ids = torch.randint(0, V + 1, (B, T), device="cuda", dtype=torch.int64)
out = torch.empty(T * B, D, device="cuda")
for _ in range(5):
    ops.gather_rows(table, ids, out, B, T, D)
torch.cuda.synchronize()
