#!/usr/bin/env python
"""The BatchNorm streaming kernels on the layer shapes of the joint step (B = 256): us per launch and TB/s of algorithmic
bytes (bn_bwd_apply: z + dy in, dz out over z = 12 B / element; bn_apply_relu: 8; bn_bwd_reduce: 8).
DS_LIB=<.so> for A/B of kernel variants, DS_STREAM_BPC=<workgroups per CU>."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import _lib, ops

if os.environ.get("DS_LIB"):
    _lib.LIB_PATH = os.environ["DS_LIB"]
SHAPES = [(802816, 64), (802816, 192), (200704, 96), (200704, 128), (200704, 288), (50176, 208), (50176, 512), (50176, 296),
          (12544, 384), (12544, 624)]


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
print("%8s %5s | %9s %6s | %9s %6s | %9s %6s" % ("M", "C", "bwd_apply", "TB/s", "apply", "TB/s", "bwd_red", "TB/s"))
for M, Cc in SHAPES:
    z = torch.randn(M, Cc, device="cuda")
    dy = torch.randn(M, Cc, device="cuda")
    y = torch.empty(M, Cc, device="cuda")
    mean, rstd, shift = torch.randn(Cc, device="cuda"), torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda")
    coef = torch.randn(2 * Cc, device="cuda") * 1e-3
    segs = ops.make_segments([(0, Cc, dy.data_ptr(), Cc)])
    dst = ops.make_segments([(0, Cc, y.data_ptr(), Cc)])
    P = ops.bn_bwd_partials(M, Cc)
    part = torch.empty(2 * Cc * P, device="cuda")
    t0 = timeit(lambda: ops.bn_bwd_apply(z, segs, M, Cc, mean, rstd, shift, coef, z))
    t1 = timeit(lambda: ops.bn_apply_relu(z, M, Cc, rstd, shift, dst))
    t2 = timeit(lambda: ops.bn_bwd_reduce(z, segs, M, Cc, mean, rstd, shift, part))
    n = M * Cc
    print("%8d %5d | %9.1f %6.2f | %9.1f %6.2f | %9.1f %6.2f" % (M, Cc, t0, 12 * n / t0 / 1e6, t1, 8 * n / t1 / 1e6, t2, 8 * n / t2 / 1e6))
    for i, t in enumerate((t0, t1, t2)):
        tot[i] += t
print("sums: bwd_apply %.1f us, apply %.1f us, bwd_reduce %.1f us" % tuple(tot))
