#!/usr/bin/env python
"""One 1x1 layer shape through the wide kernel, a few launches (for rocprofv3 --pmc passes and variant A/B):
    python scripts/wide_one.py HW K N [dgrad] [B]      DS_LIB=<alternative libds_kernels.so>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _tuning  # noqa: F401,E402  (the -DDS_TUNING library: ds_debug_* switches, DS_* knobs)
import torch
from tumblr_emotions_amd import _lib, ops

if os.environ.get("DS_LIB"):
    _lib.LIB_PATH = os.environ["DS_LIB"]
hw, K, N = (int(v) for v in sys.argv[1:4])
dgrad = len(sys.argv) > 4 and sys.argv[4] == "dgrad"
B = int(sys.argv[5]) if len(sys.argv) > 5 else 256
lib = _lib.load()
M = B * hw * hw
x = torch.randn(M, K, device="cuda")
w = torch.randn(*((N, K) if dgrad else (K, N)), device="cuda") * 0.05
z = torch.empty(M, N, device="cuda")
lib.ds_debug_conv_set_wide(2)
if dgrad:
    plan = ops.gemm_plan(M, K, N, K, N, K, transposed_w=True)
else:
    plan = ops.ConvPlan(M, 1, 1, K, K, 1, 1, 1, N, N, 0, 1, N, flags=ops.DS_EPI_STATS, pad_t=0, pad_l=0, OH=1, OW=1)
stats = torch.zeros(2 * N * max(plan.partials, 1) + 16, device="cuda")
for _ in range(3):
    plan.run(ops._p(x), ops._p(w), ops._p(z), stats=ops._p(stats))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    plan.run(ops._p(x), ops._p(w), ops._p(z), stats=ops._p(stats))
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print("%dx%d K=%d N=%d %s B=%d: %.1f us  %.1f TF/s  A = %.1f MB, z = %.1f MB" % (hw, hw, K, N, "dgrad" if dgrad else "fwd", B, us,
      2.0 * M * K * N / us / 1e6, M * K * 4 / 1e6, M * N * 4 / 1e6))
