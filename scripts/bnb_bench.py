#!/usr/bin/env python
"""BatchNorm backward on load (ds_conv_desc.bnb) per 1x1 dgrad shape of the joint step (B = 256): the separate
ds_bn_bwd_apply pass + plain wide dgrad against the dgrad that forms dz in its loader.  us per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
# (map, layer Cin, layer Cout, dy ranges): fused Branch_0/1/2 convs, Branch_3 convs, Conv2d_2b
SHAPES = [(56, 64, 64, (64,)), (28, 192, 176, (64, 160, 176)), (28, 192, 32, (32,)), (28, 256, 288, (128, 256, 288)),
          (28, 256, 64, (64,)), (14, 480, 304, (192, 288, 304)), (14, 480, 64, (64,)), (14, 512, 296, (160, 272, 296)),
          (14, 512, 280, (128, 256, 280)), (14, 512, 288, (112, 256, 288)), (14, 512, 64, (64,)), (14, 528, 448, (256, 416, 448)),
          (14, 528, 128, (128,)), (7, 832, 448, (256, 416, 448)), (7, 832, 128, (128,))]


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


tot = [0.0, 0.0, 0.0]
print("%4s %5s %5s | %9s %9s %9s | %9s" % ("HW", "Cin", "Cout", "apply", "dgrad", "apply+dgr", "bnb dgrad"))
for hw, ci, co, ends in SHAPES:
    M = B * hw * hw
    z = torch.randn(M, co, device="cuda")
    dy = torch.randn(M, co, device="cuda")
    w = torch.randn(ci, co, device="cuda") * 0.05
    dx = torch.zeros(M, ci, device="cuda")
    mean, rstd, shift = torch.randn(co, device="cuda"), torch.rand(co, device="cuda") + 0.5, torch.randn(co, device="cuda")
    coef = torch.randn(2, co, device="cuda") * 0.01
    parts, c0 = [], 0
    for c1 in ends:
        parts.append((c0, c1, dy.data_ptr() + 4 * c0, co))
        c0 = c1
    segs = ops.make_segments(parts)
    dz = torch.empty_like(z)
    plain = ops.LayerPlan(ops.DS_CONV_DGRAD, ops.DS_ARITH_F32, 0, B, hw, hw, ci, co, 1, 1, co, ci, 0)
    fused = ops.LayerPlan(ops.DS_CONV_DGRAD, ops.DS_ARITH_F32, 0, B, hw, hw, ci, co, 1, 1, co, ci, 0)
    ok = fused.enable_bn_backward_on_load(mean, rstd, shift, coef, parts)
    ta = timeit(lambda: ops.bn_bwd_apply(z, segs, M, co, mean, rstd, shift, coef, dz))
    td = timeit(lambda: plain.run(ops._p(dz), ops._p(w), ops._p(dx)))
    tf = timeit(lambda: fused.run(ops._p(z), ops._p(w), ops._p(dx))) if ok else float("nan")
    tot[0] += ta; tot[1] += td; tot[2] += tf
    print("%4d %5d %5d | %9.1f %9.1f %9.1f | %9.1f %s" % (hw, ci, co, ta, td, ta + td, tf, "" if tf < ta + td else "  (slower)"))
print("sum: apply %.1f  dgrad %.1f  apply + dgrad %.1f  bnb dgrad %.1f us" % (tot[0], tot[1], tot[0] + tot[1], tot[2]))
