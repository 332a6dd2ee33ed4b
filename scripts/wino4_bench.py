#!/usr/bin/env python
"""Winograd F(4x4,3x3) (ds_conv_wino4) against F(2x2,3x3) (ds_conv_wino) and the direct implicit GEMM on the 3x3 layer
shapes whose maps are multiples of four (B = 256): max error against the direct kernel, us per launch, TFLOP/s of the
CONVOLUTION (2*M*Cout*9*Cin / time).   usage: wino4_bench.py [B] [HWxCinxCout ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import _lib, ops

if os.environ.get("DS_LIB"):        # A/B runs of kernel variants on one box
    _lib.LIB_PATH = os.environ["DS_LIB"]

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(56, 64, 192), (28, 96, 128), (28, 128, 192), (28, 16, 32), (28, 32, 96)]
if len(sys.argv) > 2:
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in sys.argv[2:]]


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


print("%4s %5s %5s | %9s | %9s %7s | %9s %7s | %6s %9s %9s" % ("HW", "Cin", "Cout", "direct us", "F2 us", "TF", "F4 us", "TF", "F2/F4", "err z", "err stats"))
tot = [0.0, 0.0, 0.0]
for (hw, ci, co) in SHAPES:
    for dgrad in (False, True):
        kin, kout = (co, ci) if dgrad else (ci, co)
        torch.manual_seed(hw * 1000 + ci)
        x = torch.relu(torch.randn(B, hw, hw, kin, device="cuda"))
        w = torch.randn(3, 3, ci, co, device="cuda") * 0.05
        zd = torch.empty(B * hw * hw, kout, device="cuda")
        z2 = torch.empty_like(zd)
        z4 = torch.empty_like(zd)
        pivot = torch.randn(kout, device="cuda") * 0.1
        if dgrad:
            d = ops.ConvPlan(B, hw, hw, co, co, 3, 3, 1, ci, ci, ci * co, co, 1, flip=1)
        else:
            d = ops.ConvPlan(B, hw, hw, ci, ci, 3, 3, 1, co, co, ci * co, 1, co, flags=ops.DS_EPI_STATS)
        stats = torch.zeros(2 * kout * max(d.partials, 1) + 16, device="cuda")
        t_d = timeit(lambda: d.run(ops._p(x), ops._p(w), ops._p(zd), stats=ops._p(stats), pivot=ops._p(pivot)))
        if kin % 8:
            print("%4d %5d %5d | n/a (Cin %% 8)" % (hw, kin, kout))
            continue
        res = []
        for f4, z in ((False, z2), (True, z4)):
            p = ops.WinoPlan(B, hw, hw, kin, kin, kout, kout, flags=0 if dgrad else ops.DS_EPI_STATS, f4=f4)
            u = torch.empty(p.u_elems, device="cuda")
            ops.wino_transform_weights(ops._p(w), u, ci, co, dgrad, f4=f4)
            st = torch.zeros(2 * kout * max(p.partials, 1) + 16, device="cuda")
            t = timeit(lambda: p.run(ops._p(x), ops._p(u), ops._p(z), stats=ops._p(st), pivot=ops._p(pivot)))
            res.append((t, p, st))
        err = (z4 - zd).abs().max().item()
        es = 0.0
        if not dgrad:       # column statistics about the pivot against the direct kernel's
            P4 = res[1][1].partials
            s4 = res[1][2][:2 * kout * P4].view(2, kout, P4).double().sum(-1)
            sd = stats[:2 * kout * d.partials].view(2, kout, d.partials).double().sum(-1)
            es = ((s4 - sd).abs() / (sd.abs() + 1.0)).max().item()
        tot[0] += t_d
        tot[1] += res[0][0]
        tot[2] += res[1][0]
        print("%4d %5d %5d | %9.1f | %9.1f %7.1f | %9.1f %7.1f | %6.2f %9.2e %9.2e %s" % (
            hw, kin, kout, t_d, res[0][0], d.alg_flops / res[0][0] / 1e6, res[1][0], d.alg_flops / res[1][0] / 1e6,
            res[0][0] / res[1][0], err, es, "dgrad" if dgrad else ""))
print("sum: direct %.1f us, F(2x2) %.1f us, F(4x4) %.1f us" % tuple(tot))
