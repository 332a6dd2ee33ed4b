#!/usr/bin/env python
"""Conv2d_1a_7x7 at B (default 256): ds_conv_stem (fp32 MFMA), ds_conv_stem_bf16, and the path the 16-bit configurations used
before (4-channel copy + LDS-staged bf16 kernel); us per launch and GB/s of the algorithmic bytes (images once, z once)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256


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


x = torch.rand(B, 224, 224, 3, device="cuda")
w = torch.randn(7, 7, 4, 64, device="cuda") * 0.1
w[:, :, 3] = 0
nbytes = x.numel() * 4 + B * 112 * 112 * 64 * 4
for name, bf in (("ds_conv_stem", False), ("ds_conv_stem_bf16", True)):
    plan = ops.StemPlan(B, 224, 224, 4, 64, 64, bf16=bf)
    z = torch.empty(plan.M, 64, device="cuda")
    stats = torch.zeros(2, 64, plan.partials, device="cuda")
    pivot = torch.zeros(64, device="cuda")
    t = timeit(lambda: plan.run(ops._p(x), ops._p(w), ops._p(z), stats=ops._p(stats), pivot=ops._p(pivot)))
    print("%-20s %7.1f us  %5.0f GB/s" % (name, t, nbytes / t / 1e3))
x4 = torch.zeros(B, 224, 224, 4, device="cuda")
t_pad = timeit(lambda: ops.pad_channels(x, 3, x4, 4, B * 224 * 224))
plan = ops.ConvPlan(B, 224, 224, 28, 4, 7, 1, 2, 64, 64, 28 * 64, 1, 64, fold_cin=4, flags=ops.DS_EPI_STATS, dtype=ops.DS_DTYPE_BF16)
z = torch.empty(plan.M, 64, device="cuda")
stats = torch.zeros(2, 64, plan.partials, device="cuda")
t = timeit(lambda: plan.run(ops._p(x4), ops._p(w), ops._p(z), stats=ops._p(stats)))
print("%-20s %7.1f us (+ %.1f us for the 4-channel copy)" % ("staged bf16, folded", t, t_pad))

# round 6: the stem with MaxPool_2a inside (ds_conv_stem_pool) against the two launches it replaces
from tumblr_emotions_amd import _lib
lib = _lib.load()
plan = ops.StemPlan(B, 224, 224, 4, 64, 64)
z = torch.empty(plan.M, 64, device="cuda")
stats = torch.zeros(2, 64, plan.partials, device="cuda")
pivot = torch.zeros(64, device="cuda")
rstd, shift = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
y = torch.empty(B, 56, 56, 64, device="cuda")
am = torch.empty(B, 56, 56, 64, dtype=torch.uint8, device="cuda")
t0 = timeit(lambda: plan.run(ops._p(x), ops._p(w), ops._p(z), stats=ops._p(stats), pivot=ops._p(pivot)))
t1 = timeit(lambda: ops.maxpool_bn_relu_fwd(z, rstd, shift, y, am, B, 112, 112, 64, 3, 2))
P = lib.ds_conv_stem_pool_partials(B, 112, 112)
s1 = torch.zeros(2, 64, P, device="cuda")
zmax = torch.empty(B, 56, 56, 64, device="cuda")
t2 = timeit(lambda: _lib.check(lib.ds_conv_stem_pool(ops._p(x), ops._p(w), ops._p(zmax), ops._p(s1), ops._p(pivot), B, 224, 224, 4, 64,
                                                     64, ops._stream()), "stem_pool"))
print("ds_conv_stem %7.1f us + ds_maxpool_bn_relu_fwd %7.1f us = %7.1f us;  ds_conv_stem_pool %7.1f us" % (t0, t1, t0 + t1, t2))
