#!/usr/bin/env python
"""ds_conv_wino4_bf16x2 (F(4x4, 3x3) of the bf16-rounded operands, two bf16 pieces per Winograd-domain value, bf16 MFMA)
against the fp32 F(4x4) kernel and the direct bf16 kernel on the 3x3 shapes of the tower: accuracy at B = 2 against an
fp64 convolution of the bf16-rounded operands (CPU), launch times at B (default 256).

    python scripts/wino4h_bench.py [B] [--no-check]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _tuning  # noqa: F401,E402  (the -DDS_TUNING library: ds_debug_* switches, DS_* knobs)
import torch
import torch.nn.functional as F
from tumblr_emotions_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
lib = _lib.load()
SHAPES = [(56, 64, 192), (28, 96, 128), (28, 16, 32), (28, 128, 192), (28, 32, 96), (14, 96, 208), (14, 16, 48), (14, 112, 224),
          (14, 128, 256), (14, 144, 288), (14, 32, 64), (14, 160, 320), (14, 32, 128), (7, 160, 320), (7, 192, 384), (7, 48, 128)]


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


def run_h(x, u2, z, stats, pivot, Bn, hw, kin, kout, flags, y=None):
    rc = lib.ds_conv_wino4_bf16x2(ops._p(x), ops._p(u2), ops._p(z), ops._p(stats), ops._p(pivot), ops._p(y) if y is not None else None,
                                  ops.DS_DTYPE_BF16 if (y is not None and y.dtype == torch.bfloat16) else ops.DS_DTYPE_F32,
                                  Bn, hw, hw, kin, kin, kout, kout, flags, st())
    assert rc == 0, rc


def run_f(x, u, z, stats, pivot, Bn, hw, kin, kout, flags, y=None):
    rc = lib.ds_conv_wino4(ops._p(x), ops._p(u), ops._p(z), ops._p(stats), ops._p(pivot), ops._p(y) if y is not None else None,
                           Bn, hw, hw, kin, kin, kout, kout, flags, st())
    assert rc == 0, rc


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


if "--no-check" not in sys.argv:
    print("accuracy at B = 2 (fp64 convolution of the bf16-rounded operands): max|err| / max|z|, rms err / rms z")
    for (hw, ci, co) in SHAPES:
        for dgrad in (False, True):
            kin, kout = (co, ci) if dgrad else (ci, co)
            if kin % 16:
                continue
            torch.manual_seed(hw * 7 + ci)
            x = torch.relu(torch.randn(2, hw, hw, kin, device="cuda")) if not dgrad else torch.randn(2, hw, hw, kin, device="cuda") * 0.1
            w = torch.randn(3, 3, ci, co, device="cuda") * 0.05
            xr, wr = bf(x).double().cpu(), bf(w).double().cpu()
            if dgrad:
                ref = F.conv_transpose2d(xr.permute(0, 3, 1, 2), wr.permute(3, 2, 0, 1).contiguous().permute(0, 1, 2, 3), padding=1)
                # conv_transpose2d weight: [in = co, out = ci, kh, kw]
            else:
                ref = F.conv2d(xr.permute(0, 3, 1, 2), wr.permute(3, 2, 0, 1), padding=1)
            ref = ref.permute(0, 2, 3, 1).contiguous()
            z = torch.empty(2 * hw * hw, kout, device="cuda")
            zf = torch.empty_like(z)
            u2 = torch.empty(36 * ci * co, device="cuda")
            u = torch.empty(36 * ci * co, device="cuda")
            assert lib.ds_wino4_transform_weights_bf16x2(ops._p(w), ops._p(u2), ci, co, int(dgrad), st()) == 0
            assert lib.ds_wino4_transform_weights(ops._p(bf(w)), ops._p(u), ci, co, int(dgrad), st()) == 0
            P = lib.ds_conv_wino4_partials(2, hw, hw)
            stats = torch.zeros(2 * kout * P + 16, device="cuda")
            pivot = torch.zeros(kout, device="cuda")
            res = []
            for nb in (1, 2):
                lib.ds_debug_conv_wino4_set_nb(nb)
                run_h(x, u2, z, stats, pivot, 2, hw, kin, kout, 0)
                run_f(bf(x), u, zf, stats, pivot, 2, hw, kin, kout, 0)
                torch.cuda.synchronize()
                for zz in (z, zf):
                    e = zz.view(2, hw, hw, kout).double().cpu() - ref
                    res.append((float(e.abs().max() / ref.abs().max()), float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())))
            lib.ds_debug_conv_wino4_set_nb(0)
            print("%3d %4d %4d %-5s | bf16x2 NB1 %.2e %.2e  NB2 %.2e %.2e | fp32 F(4x4) of the same operands %.2e %.2e" % (
                hw, ci, co, "dgrad" if dgrad else "fwd", res[0][0], res[0][1], res[2][0], res[2][1], res[1][0], res[1][1]))
            sys.stdout.flush()

print("launch times at B = %d, us: bf16x2 F(4x4) NB = 1 / 2 / model's choice | fp32 F(4x4) | direct bf16 (fp32 x)" % B)
tot = [0.0, 0.0, 0.0]
for (hw, ci, co) in SHAPES:
    for dgrad in (False, True):
        kin, kout = (co, ci) if dgrad else (ci, co)
        if kin % 16:
            continue
        x = torch.relu(torch.randn(B, hw, hw, kin, device="cuda"))
        w = torch.randn(3, 3, ci, co, device="cuda") * 0.05
        z = torch.empty(B * hw * hw, kout, device="cuda")
        u2 = torch.empty(36 * ci * co, device="cuda")
        u = torch.empty(36 * ci * co, device="cuda")
        lib.ds_wino4_transform_weights_bf16x2(ops._p(w), ops._p(u2), ci, co, int(dgrad), st())
        lib.ds_wino4_transform_weights(ops._p(w), ops._p(u), ci, co, int(dgrad), st())
        P = lib.ds_conv_wino4_partials(B, hw, hw)
        flags = 0 if dgrad else ops.DS_EPI_STATS
        plan = ops.Bf16Plan(B, hw, hw, kin, kin, 3, 1, kout, kout, flags=flags)
        P = max(P, plan.partials)
        stats = torch.zeros(2 * kout * P + 16, device="cuda")
        pivot = torch.zeros(kout, device="cuda")
        wb = torch.empty(ops.weights_bf16_bytes(ci, co, 9, dgrad), dtype=torch.uint8, device="cuda")
        ops.weights_to_bf16(ops._p(w), wb, ci, co, 9, dgrad)
        ts = []
        for nb in (1, 2, 0):
            lib.ds_debug_conv_wino4_set_nb(nb)
            ts.append(timeit(lambda: run_h(x, u2, z, stats, pivot, B, hw, kin, kout, flags)))
        tf = timeit(lambda: run_f(x, u, z, stats, pivot, B, hw, kin, kout, flags))
        td = timeit(lambda: plan.run(ops._p(x), ops._p(wb), ops._p(z), stats=ops._p(stats), pivot=ops._p(pivot)))
        tot[0] += min(ts[0], ts[1])
        tot[1] += tf
        tot[2] += td
        print("%3d %4d %4d %-5s | %7.1f %7.1f %7.1f | %7.1f | %7.1f" % (hw, ci, co, "dgrad" if dgrad else "fwd", ts[0], ts[1], ts[2], tf, td))
        sys.stdout.flush()
print("sums: bf16x2 F(4x4) best NB %.1f us, fp32 F(4x4) %.1f us, direct bf16 %.1f us" % tuple(tot))
