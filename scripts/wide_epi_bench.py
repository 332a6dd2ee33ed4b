#!/usr/bin/env python
"""What the wide 1x1 kernel's epilogues cost: every 1x1 dgrad shape of the joint step (B = 256) with flags 0, ACCUM,
ACCUM | BNSUMS (the fused block-input dgrad's form), and the forward shapes with / without STATS."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _tuning  # noqa: F401,E402  (the -DDS_TUNING library: DS_WIDE_RING and the other knobs are honoured)
import torch
from tumblr_emotions_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
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


tot = {}
print("%4s %5s %5s %6s | %8s %8s %8s | %8s" % ("HW", "K", "N", "", "plain", "accum", "acc+sums", "ideal"))
for (hw, ci, co) in SHAPES:
    M = B * hw * hw
    for dgrad in (False, True):
        K, N = (co, ci) if dgrad else (ci, co)
        if K % 8:
            continue
        torch.manual_seed(hw * 1000 + ci + co + int(dgrad))
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(ci, co, device="cuda") * 0.05
        z = torch.zeros(M, N, device="cuda")
        y = torch.randn(M, N, device="cuda")
        res = []
        if dgrad:
            for fl in (0, ops.DS_EPI_ACCUM, ops.DS_EPI_ACCUM | ops.DS_EPI_BNSUMS):
                plan = ops.gemm_plan(M, K, N, K, N, co, transposed_w=True, flags=fl & ops.DS_EPI_ACCUM)
                P = plan.enable_bnsums(N) if fl & ops.DS_EPI_BNSUMS else 0
                stats = torch.zeros(2 * N * max(P, 1) + 16, device="cuda")
                res.append(timeit(lambda: plan.run(ops._p(x), ops._p(w), ops._p(z), stats=ops._p(stats), mask=ops._p(y))))
        else:
            for fl in (0, ops.DS_EPI_STATS):
                plan = ops.ConvPlan(M, 1, 1, K, K, 1, 1, 1, N, N, 0, 1, N, flags=fl, pad_t=0, pad_l=0, OH=1, OW=1)
                stats = torch.zeros(2 * N * max(plan.partials, 1) + 16, device="cuda")
                res.append(timeit(lambda: plan.run(ops._p(x), ops._p(w), ops._p(z), stats=ops._p(stats))))
            res.append(float("nan"))
        # fingerprint of the plain launch's output (bit patterns summed): equal across kernel variants = bit-identical
        plan0 = (ops.gemm_plan(M, K, N, K, N, co, transposed_w=True) if dgrad else
                 ops.ConvPlan(M, 1, 1, K, K, 1, 1, 1, N, N, 0, 1, N, flags=0, pad_t=0, pad_l=0, OH=1, OW=1))
        z.zero_()
        plan0.run(ops._p(x), ops._p(w), ops._p(z))
        torch.cuda.synchronize()
        fp = int(z.view(torch.int32).to(torch.int64).sum().item()) & 0xFFFFFFFF
        ideal = 2.0 * M * K * N / 157.3e6
        for i, r in enumerate(res):
            if r == r:
                tot[(dgrad, i)] = tot.get((dgrad, i), 0.0) + r
        tot[(dgrad, "ideal")] = tot.get((dgrad, "ideal"), 0.0) + ideal
        print("%4d %5d %5d %6s | %8.1f %8.1f %8.1f | %8.1f | %08x" % (hw, K, N, "dgrad" if dgrad else "", res[0], res[1], res[2], ideal, fp))
print("forward: plain %.1f  stats %.1f  ideal(157.3 TF) %.1f us" % (tot[(False, 0)], tot[(False, 1)], tot[(False, "ideal")]))
print("dgrad:   plain %.1f  accum %.1f  accum+sums %.1f  ideal %.1f us" % (tot[(True, 0)], tot[(True, 1)], tot[(True, 2)],
                                                                         tot[(True, "ideal")]))
