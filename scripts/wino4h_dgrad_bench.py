#!/usr/bin/env python
"""The 3x3 input gradients of the tower in the 16-bit configurations (dz in fp32): the LDS-staged bf16 kernel, the
register-direct bf16 kernel and ds_conv_wino4_bf16x2, us per launch at B (default 256), plain and with the BatchNorm-sums
epilogue where the kernel has one.   python scripts/wino4h_dgrad_bench.py [B]
X16=1: dz in bf16 storage (round 6: every frozen layer's dz), register-direct (x_dtype) against ds_conv_wino4_bf16x2_x16."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
lib = _lib.load()
LAYERS = [(56, 64, 192), (28, 96, 128), (28, 16, 32), (28, 128, 192), (28, 32, 96), (14, 96, 208), (14, 16, 48), (14, 112, 224),
          (14, 24, 64), (14, 128, 256), (14, 24, 64), (14, 144, 288), (14, 32, 64), (14, 160, 320), (14, 32, 128), (7, 160, 320),
          (7, 32, 128), (7, 192, 384), (7, 48, 128)]


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


def st():
    return torch.cuda.current_stream().cuda_stream


if os.environ.get("X16") == "1":
    tot = [0.0] * 4
    print("%3s %4s %4s | %8s %8s | %8s %8s | pick (with sums)" % ("HW", "Cin", "Cout", "direct", "wino-h", "dir+sums", "w-h+s16"))
    for (hw, ci, co) in LAYERS:
        K, N = co, ci
        M = B * hw * hw
        x = (torch.randn(M, K, device="cuda") * 0.1).to(torch.bfloat16)
        w = torch.randn(3, 3, ci, co, device="cuda") * 0.05
        z = torch.empty(M, N, device="cuda")
        y16 = torch.relu(torch.randn(M, N, device="cuda")).to(torch.bfloat16)
        t_d = t_ds = float("nan")
        stats = torch.zeros(2 * N * max(M // 128 + 1, lib.ds_conv_wino4_partials(B, hw, hw)) + 16, device="cuda")
        if K % 8 == 0:
            new = ops.Bf16Plan(B, hw, hw, K, K, 3, 1, N, N, flags=0)
            new.d.x_dtype = ops.DS_DTYPE_BF16
            wb = torch.empty(ops.weights_bf16_bytes(ci, co, 9, True), dtype=torch.uint8, device="cuda")
            ops.weights_to_bf16(ops._p(w), wb, ci, co, 9, True)
            t_d = timeit(lambda: new.run(ops._p(x), ops._p(wb), ops._p(z)))
            new.flags = ops.DS_EPI_BNSUMS
            new.d.ldmask, new.d.mask_dtype = N, ops.DS_DTYPE_BF16
            t_ds = timeit(lambda: new.run(ops._p(x), ops._p(wb), ops._p(z), stats=ops._p(stats), mask=ops._p(y16)))
        u2 = torch.empty(36 * ci * co, device="cuda")
        lib.ds_wino4_transform_weights_bf16x2(ops._p(w), ops._p(u2), ci, co, 1, st())

        def run_h(flags):
            rc = lib.ds_conv_wino4_bf16x2_x16(ops._p(x), ops._p(u2), ops._p(z), ops._p(stats), None, ops._p(y16) if flags else None,
                                              ops.DS_DTYPE_BF16, B, hw, hw, K, K, N, N, flags, st())
            assert rc == 0, rc
        t_h = timeit(lambda: run_h(0))
        t_hs = timeit(lambda: run_h(ops.DS_EPI_BNSUMS))
        for i, t in enumerate((t_d, t_h, t_ds, t_hs)):
            tot[i] += t
        print("%3d %4d %4d | %8.1f %8.1f | %8.1f %8.1f | %s" % (hw, ci, co, t_d, t_h, t_ds, t_hs, "direct" if t_ds < t_hs else "wino-h"))
        sys.stdout.flush()
    print("sums: direct %.0f wino-h %.0f | direct + sums %.0f wino-h + sums %.0f" % tuple(tot))
    sys.exit(0)

tot = [0.0] * 5
print("%3s %4s %4s | %8s %8s %8s | %8s %8s | pick" % ("HW", "Cin", "Cout", "staged", "direct", "wino-h", "dir+sums", "w-h+s16"))
for (hw, ci, co) in LAYERS:
    K, N = co, ci
    M = B * hw * hw
    x = torch.randn(M, K, device="cuda") * 0.1
    w = torch.randn(3, 3, ci, co, device="cuda") * 0.05
    z = torch.empty(M, N, device="cuda")
    y = torch.relu(torch.randn(M, N, device="cuda"))
    y16 = y.to(torch.bfloat16)
    old = ops.ConvPlan(B, hw, hw, K, K, 3, 3, 1, N, N, ci * co, co, 1, flip=1, dtype=ops.DS_DTYPE_BF16)
    t_st = timeit(lambda: old.run(ops._p(x), ops._p(w), ops._p(z)))
    t_d = t_ds = float("nan")
    if K % 8 == 0:
        new = ops.Bf16Plan(B, hw, hw, K, K, 3, 1, N, N, flags=0)
        wb = torch.empty(ops.weights_bf16_bytes(ci, co, 9, True), dtype=torch.uint8, device="cuda")
        ops.weights_to_bf16(ops._p(w), wb, ci, co, 9, True)
        stats = torch.zeros(2 * N * max(lib.ds_conv_bf16_partials(new.d), lib.ds_conv_wino4_partials(B, hw, hw)) + 16, device="cuda")
        t_d = timeit(lambda: new.run(ops._p(x), ops._p(wb), ops._p(z)))
        new.flags = ops.DS_EPI_BNSUMS
        new.d.ldmask, new.d.mask_dtype = N, ops.DS_DTYPE_BF16
        t_ds = timeit(lambda: new.run(ops._p(x), ops._p(wb), ops._p(z), stats=ops._p(stats), mask=ops._p(y16)))
    u2 = torch.empty(36 * ci * co, device="cuda")
    lib.ds_wino4_transform_weights_bf16x2(ops._p(w), ops._p(u2), ci, co, 1, st())

    def run_h(flags, yy=None):
        rc = lib.ds_conv_wino4_bf16x2(ops._p(x), ops._p(u2), ops._p(z), ops._p(stats), None, ops._p(yy) if flags else None,
                                      ops.DS_DTYPE_BF16 if (yy is not None and yy.dtype == torch.bfloat16) else ops.DS_DTYPE_F32,
                                      B, hw, hw, K, K, N, N, flags, st())
        assert rc == 0, rc
    t_h = timeit(lambda: run_h(0))
    t_hs = timeit(lambda: run_h(ops.DS_EPI_BNSUMS, y16))
    best = min(t_st, t_d, t_h)
    for i, t in enumerate((t_st, t_d, t_h, t_ds, t_hs)):
        tot[i] += t
    print("%3d %4d %4d | %8.1f %8.1f %8.1f | %8.1f %8.1f | %s" % (hw, ci, co, t_st, t_d, t_h, t_ds, t_hs,
                                                                  "staged" if best == t_st else ("direct" if best == t_d else "wino-h")))
    sys.stdout.flush()
print("sums: staged %.0f direct %.0f wino-h %.0f | direct + sums %.0f wino-h + sums %.0f" % tuple(tot))
