#!/usr/bin/env python
"""Wide-tile register-direct 1x1 kernel against the LDS-tile kernels on the 1x1 layer shapes of the joint step
(B = 256): forward (fused Branch_0/1/2 1x1, Branch_3 1x1, Conv2d_2b) with statistics, and their dgrads."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _tuning  # noqa: F401,E402  (the -DDS_TUNING library: ds_debug_* switches, DS_* knobs)
import torch
from tumblr_emotions_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
# (HW, Cin, Cout) forward shapes
SHAPES = [(56, 64, 64), (28, 192, 176), (28, 192, 32), (28, 256, 288), (28, 256, 64), (14, 480, 304), (14, 480, 64),
          (14, 512, 296), (14, 512, 280), (14, 512, 288), (14, 512, 64), (14, 528, 448), (14, 528, 128),
          (7, 832, 448), (7, 832, 128), (7, 832, 624)]


def timeit(f, reps=10):
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = [0.0, 0.0, 0.0, 0.0]
print("%4s %5s %5s %6s | %9s %7s | %9s %7s | %6s" % ("HW", "K", "N", "", "lds us", "TF", "wide us", "TF", "ratio"))
for (hw, ci, co) in SHAPES:
    M = B * hw * hw
    for dgrad in (False, True):
        K, N = (co, ci) if dgrad else (ci, co)
        if K % 8:
            continue
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(ci, co, device="cuda") * 0.05
        z = torch.empty(M, N, device="cuda")
        res = []
        for mode in (0, 2, 1):
            lib.ds_debug_conv_set_wide(mode)
            if dgrad:
                plan = ops.gemm_plan(M, K, N, K, N, co, transposed_w=True)
            else:
                plan = ops.ConvPlan(M, 1, 1, K, K, 1, 1, 1, N, N, 0, 1, N, flags=ops.DS_EPI_STATS, pad_t=0, pad_l=0, OH=1, OW=1)
            stats = torch.zeros(2 * N * max(plan.partials, 1) + 16, device="cuda")
            res.append(timeit(lambda: plan.run(ops._p(x), ops._p(w), ops._p(z), stats=ops._p(stats))))
        fl = 2.0 * M * K * N
        lib.ds_debug_conv_set_wide(1)
        auto = ops.ConvPlan(M, 1, 1, K, K, 1, 1, 1, N, N, 0, 1, N, pad_t=0, pad_l=0, OH=1, OW=1) if not dgrad else None
        tot[0] += res[0]
        tot[1] += res[1]
        tot[2] += min(res[:2])
        tot[3] += res[2]
        print("%4d %5d %5d %6s | %9.1f %7.1f | %9.1f %7.1f | %6.2f" % (hw, K, N, "dgrad" if dgrad else "", res[0], fl / res[0] / 1e6,
                                                                     res[1], fl / res[1] / 1e6, res[0] / res[1]))
lib.ds_debug_conv_set_wide(1)
print("sum: lds %.1f us, wide %.1f us, best of both %.1f us, library's own choice %.1f us" % tuple(tot))
