#!/usr/bin/env python
"""ds_conv_plan's split-K choice for the F(4x4) launches of the 14x14 / 7x7 layers at small batches: us per ds_conv_run with the
statistics (forward) / BatchNorm-sums (dgrad) epilogue, unsplit against the model's slice count and forced 2 / 3 / 4 slices
(DS_WINO4_SPLITK, tuning library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = [(14, 96, 208), (14, 128, 256), (14, 160, 320), (7, 160, 320), (7, 192, 384), (28, 128, 192)]


def timeit(f, reps=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for hw, ci, co in SHAPES:
    for role in (ops.DS_CONV_FWD, ops.DS_CONV_DGRAD):
        dg = role == ops.DS_CONV_DGRAD
        kin, kout = (co, ci) if dg else (ci, co)
        x = torch.randn(B * hw * hw, kin, device="cuda")
        w = torch.randn(3, 3, ci, co, device="cuda") * 0.05
        z = torch.empty(B * hw * hw, kout, device="cuda")
        y = torch.relu(torch.randn(B * hw * hw, kout, device="cuda"))
        pivot = torch.zeros(kout, device="cuda")
        out = []
        for opt in (ops.DS_PLAN_NO_SPLITK, 0):
            pl = ops.LayerPlan(role, ops.DS_ARITH_F32, opt, B, hw, hw, ci, co, 3, 1, kin, kout, 0 if dg else ops.DS_EPI_STATS)
            if pl.family != ops.DS_FAM_WINO4:
                out.append((0, 0.0))
                continue
            pl.alloc_weights(x.device)
            pl.prepare(ops._p(w))
            if dg:
                pl.enable_bnsums(kout)
            st = torch.zeros(2 * kout * max(pl.partials, 1) + 16, device="cuda")
            if pl.ws_bytes:
                pl.set_workspace(torch.empty(pl.ws_bytes // 4, device="cuda"))
            t = timeit(lambda: pl.run(ops._p(x), ops._p(w), ops._p(z), stats=ops._p(st), pivot=None if dg else ops._p(pivot), mask=ops._p(y) if dg else None))
            out.append((pl.splitk, t))
        print("B=%3d %2dx%-2d %4d -> %4d %-5s | unsplit %7.1f us | splits %d: %7.1f us" % (B, hw, hw, kin, kout, "dgrad" if dg else "fwd", out[0][1], out[1][0], out[1][1]))
