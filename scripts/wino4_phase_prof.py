#!/usr/bin/env python
"""Where does a workgroup of conv_wino4_kernel spend its time?  Phase table per instantiation.

A -DDS_W4_PROF build of csrc/conv_wino4.hip (the library build compiles none of it) stamps s_memtime in wave 0 of every
workgroup at its phase boundaries; this script builds that variant when it is missing (and, with --abl, variants with
pieces of the K loop compiled out: DS_W4_ABL bit 1 weight-fragment loads, 2 pixel loads, 4 transform VALU, 8 LDS writes
of V), runs the 3x3 layer shapes of the tower and prints per shape: launch time, workgroups and rounds, and the mean over
workgroups of  prologue (index math, first pixel + weight requests, first transform) | first barrier | K loop (and per
16-channel K step) | per channel block: park | gather + output transform + stores issued | statistics | store drain.

    python scripts/wino4_phase_prof.py [B] [--abl] [--lib path.so] [--def NAME=VALUE ...] [--build-only] [HWxCinxCout[d][s] ...]
        d = dgrad (reduction over Cout), s = BatchNorm-sums epilogue (DS_EPI_BNSUMS, dgrad only)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tumblr_emotions_amd import _lib, ops

CSRC = os.path.join(ROOT, "tumblr_emotions_amd", "csrc")
MB = os.path.join(ROOT, "scripts", "microbench")


def build(abl=0, extra=()):
    tag = "".join("_" + e[2:].replace("=", "") for e in extra if e.startswith("-D"))
    so = os.path.join(MB, "libw4prof%s%s.so" % ("_abl%d" % abl if abl else "", tag))
    src = os.path.join(CSRC, "conv_wino4.hip")
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src):
        return so
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, "-Wno-unused-result", "-DDS_W4_PROF", "-DDS_TUNING", "-DDS_W4_ABL=%d" % abl, "-shared", src,
           os.path.join(CSRC, "error.cpp"), "-o", so] + list(extra)
    subprocess.run(cmd, check=True)
    return so


def load(path):
    lib = C.CDLL(path)
    for name in ("ds_conv_wino4", "ds_wino4_transform_weights", "ds_conv_wino4_partials", "ds_conv_wino4_prefer",
                 "ds_debug_conv_wino4_set_nb"):
        res, args = _lib.SIGNATURES[name]
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    lib.ds_debug_w4_set_prof.restype, lib.ds_debug_w4_set_prof.argtypes = None, [C.c_void_p]
    return lib


def parse(s):
    flags = ""
    while s and s[-1] in "ds":
        flags, s = s[-1] + flags, s[:-1]
    hw, ci, co = (int(v) for v in s.split("x"))
    return hw, ci, co, "d" in flags, "s" in flags


def run_shape(lib, B, hw, ci, co, dgrad, bns, reps=20):
    kin, kout = (co, ci) if dgrad else (ci, co)
    torch.manual_seed(hw * 1000 + ci)
    x = torch.relu(torch.randn(B, hw, hw, kin, device="cuda"))
    w = torch.randn(3, 3, ci, co, device="cuda") * 0.05
    z = torch.empty(B * hw * hw, kout, device="cuda")
    y = torch.relu(torch.randn(B * hw * hw, kout, device="cuda")) if bns else None
    pivot = torch.zeros(kout, device="cuda")
    u = torch.empty(36 * ci * co, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.ds_wino4_transform_weights(ops._p(w), ops._p(u), ci, co, int(dgrad), st) == 0
    P = lib.ds_conv_wino4_partials(B, hw, hw)
    stats = torch.zeros(2 * kout * P + 16, device="cuda")
    flags = ops.DS_EPI_BNSUMS if bns else (0 if dgrad else ops.DS_EPI_STATS)
    groups = (B * ((hw + 3) // 4) ** 2 + 31) // 32
    prof = torch.zeros((groups * ((kout + 31) // 32) + 7) // 8 * 8, 16, dtype=torch.int64, device="cuda")

    def launch():
        rc = lib.ds_conv_wino4(ops._p(x), ops._p(u), ops._p(z), ops._p(stats), ops._p(pivot), ops._p(y) if bns else None,
                               B, hw, hw, kin, kin, kout, kout, flags, st)
        assert rc == 0, rc

    lib.ds_debug_w4_set_prof(None)
    for _ in range(2):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us_plain = e0.elapsed_time(e1) / reps * 1e3
    lib.ds_debug_w4_set_prof(prof.data_ptr())
    e0.record()
    launch()
    e1.record()
    torch.cuda.synchronize()
    us_prof = e0.elapsed_time(e1) * 1e3
    lib.ds_debug_w4_set_prof(None)
    p = prof.cpu().numpy().astype(np.float64)
    p = p[p[:, 0] > 0]                      # workgroups that ran (surplus ones return before the first stamp is stored)
    # s_memtime counts shader cycles and every CU / XCD has its own epoch: only differences INSIDE a workgroup mean anything.
    # Calibration: s_memrealtime (100 MHz, one epoch for the device) stamped at the start and the end of every workgroup.
    tick = float(np.sum(p[:, 13] - p[:, 12]) / 100.0 / np.sum(p[:, 10] - p[:, 0]))
    xcc = p[:, 11].astype(np.int64)
    span_us = (p[:, 13].max() - p[:, 12].min()) / 100.0          # first workgroup's start to the last one's end
    two = bool((p[:, 7] > 0).any())
    ks = kin // 16
    d = lambda a, b: float(np.mean(p[:, b] - p[:, a]) * tick)
    row = dict(us=us_plain, us_prof=us_prof, wgs=len(p), rounds=len(p) / 256.0, nb=2 if two else 1, ksteps=ks,
               mhz=1.0 / tick, prologue=d(0, 1), barrier0=d(1, 2), kloop=d(2, 3), kstep=d(2, 3) / ks,
               park0=d(3, 4), xform0=d(4, 5), stats0=d(5, 6))
    last = 6
    if two:
        row.update(park1=d(6, 7), xform1=d(7, 8), stats1=d(8, 9))
        last = 9
    row.update(drain=d(last, 10), total=d(0, 10))
    # how evenly are the workgroups spread over the XCDs, and how long does the chip wait for the last one?
    # busy fraction of the CUs between the first start and the last end (dispatch gaps, partial last round)
    row["busy"] = float(np.sum(p[:, 13] - p[:, 12]) / 100.0 / (256.0 * span_us))
    row["span"] = span_us
    row["xcds"] = len(np.unique(xcc))
    row["mfma_us"] = ks * 72 * row["nb"] * 64 / 2400.0          # 72 NB MFMAs of 64 cycles per K step at 2.4 GHz
    return row


def main():
    args = [a for a in sys.argv[1:]]
    abl = "--abl" in args
    args = [a for a in args if a != "--abl"]
    defs = []
    while "--def" in args:
        i = args.index("--def")
        defs.append("-D" + args[i + 1])
        del args[i:i + 2]
    if "--build-only" in args:
        print(build(0, defs))
        return
    libpath = None
    if "--lib" in args:
        i = args.index("--lib")
        libpath = args[i + 1]
        del args[i:i + 2]
    B = int(args[0]) if args and args[0].isdigit() else 256
    shapes = [a for a in args if "x" in a] or ["56x64x192", "56x64x192ds", "28x96x128", "28x96x128ds", "28x128x192", "28x128x192ds",
                                               "28x32x96", "28x32x96ds", "28x16x32", "14x96x208", "14x96x208ds", "14x160x320",
                                               "14x160x320ds", "14x32x128ds", "7x160x320", "7x192x384ds"]
    variants = [(0, "full kernel")]
    if abl:
        variants += [(1, "no weight-fragment loads in the loop"), (2, "no pixel loads in the loop"), (4, "no transform VALU"),
                     (12, "no transform VALU, no LDS writes"), (15, "MFMAs + fragment reads only")]
    for a, name in variants:
        lib = load(libpath if (libpath and a == 0) else build(a, defs))
        print("== %s (B = %d) %s ==" % (name, B, " ".join(defs)))
        print("%-14s %2s %7s %6s %6s | %5s %5s %6s %6s | %5s %6s %5s | %5s %6s %5s | %5s %6s %5s | %5s %5s" % (
            "shape", "NB", "us", "wgs", "rounds", "prol", "bar0", "Kloop", "/step", "park0", "xform0", "stat0", "park1", "xform1",
            "stat1", "drain", "total", "mfma", "busy", "MHz"))
        for sname in shapes:
            hw, ci, co, dgrad, bns = parse(sname)
            r = run_shape(lib, B, hw, ci, co, dgrad, bns)
            print("%-14s %2d %7.1f %6d %6.2f | %5.2f %5.2f %6.2f %6.2f | %5.2f %6.2f %5.2f | %5.2f %6.2f %5.2f | %5.2f %6.2f %5.2f | %5.2f %5.0f" % (
                sname, r["nb"], r["us"], r["wgs"], r["rounds"], r["prologue"], r["barrier0"], r["kloop"], r["kstep"], r["park0"],
                r["xform0"], r["stats0"], r.get("park1", 0), r.get("xform1", 0), r.get("stats1", 0), r["drain"], r["total"],
                r["mfma_us"], r["busy"], r["mhz"]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
