#!/usr/bin/env python
"""ds_conv_fp8 against ds_conv_bf16 per conv shape of the tower (forward with statistics from 16-bit activation storage,
dgrad from fp32 dz), at cfg5's per-GPU share (B = 128) unless another batch is given: which layers fp8 wins.  us."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(56, 64, 64, 1), (56, 64, 192, 3), (28, 192, 176, 1), (28, 192, 32, 1), (28, 96, 128, 3), (28, 16, 32, 3), (28, 256, 288, 1),
          (28, 256, 64, 1), (28, 128, 192, 3), (28, 32, 96, 3), (14, 480, 304, 1), (14, 480, 64, 1), (14, 96, 208, 3), (14, 16, 48, 3),
          (14, 512, 296, 1), (14, 112, 224, 3), (14, 24, 64, 3), (14, 512, 280, 1), (14, 128, 256, 3), (14, 512, 288, 1),
          (14, 144, 288, 3), (14, 32, 64, 3), (14, 528, 448, 1), (14, 528, 128, 1), (14, 160, 320, 3), (14, 32, 128, 3),
          (7, 832, 448, 1), (7, 832, 128, 1), (7, 160, 320, 3), (7, 832, 624, 1), (7, 192, 384, 3), (7, 48, 128, 3)]


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


tot = {}
print("%4s %5s %5s %2s %6s | %8s %8s | %6s" % ("HW", "Cin", "Cout", "k", "", "bf16 us", "fp8 us", "bf16/fp8"))
for (hw, ci, co, k) in SHAPES:
    for dgrad in (False, True):
        role = ops.DS_CONV_DGRAD if dgrad else ops.DS_CONV_FWD
        K, N = (co, ci) if dgrad else (ci, co)
        M = B * hw * hw
        x = torch.randn(M, K, device="cuda")
        if not dgrad:
            x = x.to(torch.bfloat16)                      # 16-bit activation storage (the fp8 / bf16 configurations' default)
        w = torch.randn(k, k, ci, co, device="cuda") * 0.05
        z = torch.empty(M, N, device="cuda")
        res = []
        for arith in (ops.DS_ARITH_BF16, ops.DS_ARITH_FP8):
            plan = ops.LayerPlan(role, arith, ops.DS_PLAN_ACT16, B, hw, hw, ci, co, k, 1, K, N, 0 if dgrad else ops.DS_EPI_STATS)
            want = ops.DS_FAM_BF16D if arith == ops.DS_ARITH_BF16 else ops.DS_FAM_FP8D
            if plan.family != want:
                res.append(float("nan"))
                continue
            plan.alloc_weights("cuda")
            plan.prepare(ops._p(w))
            plan.d.x_dtype = ops.act_dtype(x)
            stats = torch.zeros(2 * N * max(plan.partials, 1) + 16, device="cuda")
            amax = torch.zeros(ops.AMAX_FLOATS, device="cuda")
            if arith == ops.DS_ARITH_FP8:
                ops.absmax(x, M * K, amax)
            res.append(timeit(lambda: plan.run(ops._p(x), ops._p(w), ops._p(z), stats=ops._p(stats), x_amax=ops._p(amax))))
        key = "dgrad" if dgrad else "fwd"
        if res[0] == res[0] and res[1] == res[1]:
            t = tot.setdefault(key, [0.0, 0.0, 0.0])
            t[0] += res[0]; t[1] += res[1]; t[2] += min(res)
        print("%4d %5d %5d %2d %6s | %8.1f %8.1f | %6.2f%s" % (hw, ci, co, k, "dgrad" if dgrad else "", res[0], res[1], res[0] / res[1],
                                                               "   fp8 wins" if res[1] < res[0] else ""))
for key, t in tot.items():
    print("%s: bf16 %.1f us, fp8 %.1f us, best of both %.1f us" % (key, *t))
