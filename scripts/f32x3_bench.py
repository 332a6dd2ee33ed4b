#!/usr/bin/env python
"""ds_conv_f32x3 (fp32 products on the bf16 matrix cores, three bf16 pieces per operand) against the fp32 kernels on the
1x1 layer shapes of the joint step (B = 256): error of both against an fp64 product on a row sample, us per launch.
    python scripts/f32x3_bench.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(56, 64, 64), (28, 192, 176), (28, 192, 32), (28, 256, 288), (28, 256, 64), (14, 480, 304), (14, 480, 64),
          (14, 512, 296), (14, 512, 64), (14, 528, 448), (14, 528, 128), (7, 832, 448), (7, 832, 128), (7, 832, 624)]


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


print("%4s %5s %5s | %9s %7s %9s | %9s %7s %9s | %6s" % ("HW", "K", "N", "fp32 us", "TF", "rel err", "f32x3 us", "TF", "rel err", "ratio"))
tot = [0.0, 0.0]
for hw, K, N in SHAPES:
    M = B * hw * hw
    torch.manual_seed(K * 1000 + N)
    x = torch.relu(torch.randn(M, K, device="cuda"))
    w = torch.randn(K, N, device="cuda") * 0.05
    pivot = torch.zeros(N, device="cuda")
    rows = torch.randint(0, M, (512,), device="cuda")
    ref = x[rows].double() @ w.double()
    out = []
    p = ops.ConvPlan(M, 1, 1, K, K, 1, 1, 1, N, N, 0, 1, N, flags=ops.DS_EPI_STATS, pad_t=0, pad_l=0, OH=1, OW=1)
    z = torch.empty(M, N, device="cuda")
    st = torch.zeros(2 * N * max(p.partials, 1) + 16, device="cuda")
    t0 = timeit(lambda: p.run(ops._p(x), ops._p(w), ops._p(z), stats=ops._p(st), pivot=ops._p(pivot)))
    e0 = float(((z[rows].double() - ref).norm() / ref.norm()))
    q = ops.F32x3Plan(M, 1, 1, K, K, 1, 1, N, N, flags=ops.DS_EPI_STATS, pad_t=0, pad_l=0, OH=1, OW=1)
    wb = torch.empty(ops.weights_f32x3_bytes(K, N, 1, False), dtype=torch.uint8, device="cuda")
    ops.weights_to_f32x3(ops._p(w), wb, K, N, 1, False)
    z3 = torch.empty(M, N, device="cuda")
    st3 = torch.zeros(2 * N * max(q.partials, 1) + 16, device="cuda")
    t1 = timeit(lambda: q.run(ops._p(x), ops._p(wb), ops._p(z3), stats=ops._p(st3), pivot=ops._p(pivot)))
    e1 = float(((z3[rows].double() - ref).norm() / ref.norm()))
    fl = 2.0 * M * K * N
    tot[0] += t0
    tot[1] += t1
    print("%4d %5d %5d | %9.1f %7.1f %9.2e | %9.1f %7.1f %9.2e | %6.2f" % (hw, K, N, t0, fl / t0 / 1e6, e0, t1, fl / t1 / 1e6, e1, t0 / t1))
print("sum: fp32 %.1f us, f32x3 %.1f us" % tuple(tot))
