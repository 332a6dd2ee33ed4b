#!/usr/bin/env python
"""ds_conv_bf16 (register-direct bf16) against the LDS-staged bf16 kernel and the fp32 paths on the conv shapes of
the joint step (B = 256): forward with statistics and dgrad."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
# (HW, Cin, Cout, k)
SHAPES = [(56, 64, 64, 1), (56, 64, 192, 3), (28, 192, 176, 1), (28, 96, 128, 3), (28, 16, 32, 3), (28, 256, 288, 1),
          (28, 128, 192, 3), (28, 32, 96, 3), (14, 480, 304, 1), (14, 96, 208, 3), (14, 16, 48, 3), (14, 512, 296, 1),
          (14, 112, 224, 3), (14, 24, 64, 3), (14, 528, 448, 1), (14, 160, 320, 3), (14, 32, 128, 3), (7, 832, 448, 1),
          (7, 160, 320, 3), (7, 832, 624, 1), (7, 192, 384, 3), (7, 48, 128, 3)]


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


tot = [0.0, 0.0]
print("%4s %5s %5s %2s %6s | %9s %7s | %9s %7s | %6s" % ("HW", "Cin", "Cout", "k", "", "staged us", "TF", "direct us", "TF", "ratio"))
for (hw, ci, co, k) in SHAPES:
    for dgrad in (False, True):
        K, N = (co, ci) if dgrad else (ci, co)
        if K % 8:
            continue
        x = torch.randn(B * hw * hw, K, device="cuda")
        w = torch.randn(k, k, ci, co, device="cuda") * 0.05
        z = torch.empty(B * hw * hw, N, device="cuda")
        if dgrad:
            old = ops.ConvPlan(B, hw, hw, K, K, k, k, 1, N, N, ci * co, co, 1, flip=1, dtype=ops.DS_DTYPE_BF16)
        else:
            old = ops.ConvPlan(B, hw, hw, K, K, k, k, 1, N, N, ci * co, 1, co, flags=ops.DS_EPI_STATS, dtype=ops.DS_DTYPE_BF16)
        new = ops.Bf16Plan(B, hw, hw, K, K, k, 1, N, N, flags=0 if dgrad else ops.DS_EPI_STATS)
        wb = torch.empty(ops.weights_bf16_bytes(ci, co, k * k, dgrad), dtype=torch.uint8, device="cuda")
        ops.weights_to_bf16(ops._p(w), wb, ci, co, k * k, dgrad)
        stats = torch.zeros(2 * N * max(old.partials, new.partials, 1) + 16, device="cuda")
        t0 = timeit(lambda: old.run(ops._p(x), ops._p(w), ops._p(z), stats=ops._p(stats)))
        t1 = timeit(lambda: new.run(ops._p(x), ops._p(wb), ops._p(z), stats=ops._p(stats)))
        fl = new.alg_flops
        tot[0] += t0
        tot[1] += t1
        print("%4d %5d %5d %2d %6s | %9.1f %7.1f | %9.1f %7.1f | %6.2f" % (hw, ci, co, k, "dgrad" if dgrad else "", t0, fl / t0 / 1e6,
                                                                         t1, fl / t1 / 1e6, t0 / t1))
print("sum: staged %.1f us, direct %.1f us" % tuple(tot))
