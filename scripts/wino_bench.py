#!/usr/bin/env python
"""Winograd F(2x2,3x3) kernel against the direct implicit-GEMM kernels on the 3x3 layer shapes of the joint step
(B = 256): us per launch and TFLOP/s of the CONVOLUTION (2*M*Cout*9*Cin / time) for both."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import _lib, ops

if os.environ.get("DS_LIB"):        # A/B runs of kernel variants on one box
    _lib.LIB_PATH = os.environ["DS_LIB"]

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(56, 64, 192), (28, 96, 128), (28, 128, 192), (28, 16, 32), (28, 32, 96), (14, 96, 208), (14, 112, 224),
          (14, 128, 256), (14, 144, 288), (14, 160, 320), (14, 16, 48), (14, 24, 64), (14, 32, 64), (14, 32, 128),
          (7, 160, 320), (7, 192, 384), (7, 32, 128), (7, 48, 128)]


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


print("%4s %5s %5s | %9s %8s | %9s %8s | %6s   (forward; then dgrad)" % ("HW", "Cin", "Cout", "direct us", "TF", "wino us", "TF", "speedup"))
tot = [0.0, 0.0]
for (hw, ci, co) in SHAPES:
    for dgrad in (False, True):
        kin, kout = (co, ci) if dgrad else (ci, co)
        x = torch.randn(B, hw, hw, kin, device="cuda")
        w = torch.randn(3, 3, ci, co, device="cuda") * 0.05
        z = torch.empty(B * hw * hw, kout, device="cuda")
        if dgrad:
            d = ops.ConvPlan(B, hw, hw, co, co, 3, 3, 1, ci, ci, ci * co, co, 1, flip=1)
        else:
            d = ops.ConvPlan(B, hw, hw, ci, ci, 3, 3, 1, co, co, ci * co, 1, co, flags=ops.DS_EPI_STATS)
        stats = torch.zeros(2 * kout * max(d.partials, 1) + 16, device="cuda")
        t_d = timeit(lambda: d.run(ops._p(x), ops._p(w), ops._p(z), stats=ops._p(stats)))
        u = torch.empty(16, kout, kin, device="cuda")
        ops.wino_transform_weights(ops._p(w), u, ci, co, dgrad)
        if kin % 8:
            print("%4d %5d %5d | %9.1f %8.1f | %9s" % (hw, kin, kout, t_d, d.alg_flops / t_d / 1e6, "n/a (Cin % 8)"))
            continue
        p = ops.WinoPlan(B, hw, hw, kin, kin, kout, kout, flags=0 if dgrad else ops.DS_EPI_STATS)
        stats2 = torch.zeros(2 * kout * max(p.partials, 1) + 16, device="cuda")
        t_w = timeit(lambda: p.run(ops._p(x), ops._p(u), ops._p(z), stats=ops._p(stats2)))
        tot[0] += t_d
        tot[1] += t_w
        print("%4d %5d %5d | %9.1f %8.1f | %9.1f %8.1f | %6.2f %s" % (hw, kin, kout, t_d, d.alg_flops / t_d / 1e6, t_w,
                                                                      p.alg_flops / t_w / 1e6, t_d / t_w, "dgrad" if dgrad else ""))
print("sum of the layers both can run: direct %.1f us, winograd %.1f us" % tuple(tot))
