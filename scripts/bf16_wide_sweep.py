#!/usr/bin/env python
"""ds_conv_bf16 on the 1x1 shapes of the tower: time and algorithmic HBM rate against the column blocks per wave.

The bf16 1x1 launches are streaming kernels (a 32x32x16 bf16 MFMA costs 1/16 of the fp32 one): the question is how many
bytes per second a launch moves, not how many FLOP/s.  Per shape and epilogue: us at max NB = 8 (the shipped choice), 4,
3, 2, and GB/s of the algorithmic bytes (x once, z once, + accumulate / activation reads) at the best.

    python scripts/bf16_wide_sweep.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _tuning  # noqa: F401,E402  (the -DDS_TUNING library: ds_debug_* switches, DS_* knobs)
import torch
from tumblr_emotions_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
lib = _lib.load()
# fused 1x1 of every Mixed block (Cin -> b0 + b1a + b2a), two Branch_3 1x1, Conv2d_2b
SHAPES = [(56, 64, 64), (28, 192, 176), (28, 256, 288), (14, 480, 304), (14, 512, 296), (14, 512, 288), (14, 528, 448),
          (7, 832, 448), (7, 832, 624), (28, 256, 64), (14, 512, 64), (7, 832, 128)]
NBS = [8, 4, 3, 2]


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


sums = {}
print("%3s %4s %4s %-22s | %s | %6s %5s" % ("HW", "Cin", "Cout", "launch", " ".join("%7s" % ("NB<=%d" % n) for n in NBS), "GB/s", "best"))
for (hw, ci, co) in SHAPES:
    M = B * hw * hw
    for name, dgrad, x16, flags in (("fwd x16 +stats", False, True, ops.DS_EPI_STATS), ("fwd x32 +stats", False, False, ops.DS_EPI_STATS),
                                    ("dgrad", True, False, 0), ("dgrad +acc +sums y16", True, False, ops.DS_EPI_ACCUM | ops.DS_EPI_BNSUMS)):
        K, N = (co, ci) if dgrad else (ci, co)
        if K % 8 or (x16 and K % 8):
            continue
        x = torch.randn(M, K, device="cuda")
        if x16:
            x = x.to(torch.bfloat16)
        w = torch.randn(1, 1, ci, co, device="cuda") * 0.05
        z = torch.zeros(M, N, device="cuda")
        y = torch.relu(torch.randn(M, N, device="cuda")).to(torch.bfloat16) if flags & ops.DS_EPI_BNSUMS else None
        plan = ops.Bf16Plan(B, hw, hw, K, K, 1, 1, N, N, flags=flags)
        plan.d.x_dtype = ops.DS_DTYPE_BF16 if x16 else ops.DS_DTYPE_F32
        if y is not None:
            plan.d.ldmask, plan.d.mask_dtype = N, ops.DS_DTYPE_BF16
        wb = torch.empty(ops.weights_bf16_bytes(ci, co, 1, dgrad), dtype=torch.uint8, device="cuda")
        ops.weights_to_bf16(ops._p(w), wb, ci, co, 1, dgrad)
        P = lib.ds_conv_bf16_partials(plan.d)
        stats = torch.zeros(2 * N * P + 16, device="cuda")
        pivot = torch.zeros(N, device="cuda")
        nbytes = M * K * (2 if x16 else 4) + M * N * 4
        if flags & ops.DS_EPI_ACCUM:
            nbytes += M * N * 4
        if flags & ops.DS_EPI_BNSUMS:
            nbytes += M * N * 2
        ts = []
        for nb in NBS:
            lib.ds_debug_conv_bf16_set_max_nb(nb)
            ts.append(timeit(lambda: plan.run(ops._p(x), ops._p(wb), ops._p(z), stats=ops._p(stats), pivot=ops._p(pivot),
                                              mask=ops._p(y) if y is not None else None)))
        lib.ds_debug_conv_bf16_set_max_nb(8)
        best = min(range(len(NBS)), key=lambda i: ts[i])
        for i, nb in enumerate(NBS):
            sums[(name, nb)] = sums.get((name, nb), 0.0) + ts[i]
        sums[(name, "best")] = sums.get((name, "best"), 0.0) + ts[best]
        sums[(name, "floor")] = sums.get((name, "floor"), 0.0) + nbytes / 6.3e6
        print("%3d %4d %4d %-22s | %s | %6.0f %5d" % (hw, ci, co, name, " ".join("%7.1f" % t for t in ts), nbytes / ts[best] / 1e3, NBS[best]))
print()
for name in ("fwd x16 +stats", "fwd x32 +stats", "dgrad", "dgrad +acc +sums y16"):
    print("%-22s sum us: %s | best per shape %.1f | bytes / 6.3 TB/s %.1f" % (
        name, " ".join("NB<=%d %.1f" % (nb, sums[(name, nb)]) for nb in NBS), sums[(name, "best")], sums[(name, "floor")]))
