#!/usr/bin/env python
"""Phase stamps of one mid-grid ds_conv_wino4 workgroup and the launch time.  Needs a build with the s_memrealtime
stamps and ds_debug_wino4_prof (commit 3b6834c's csrc/conv_wino4.hip compiled with -DDS_W4_PROF, optionally the
-DDS_W4_X_* ablation switches; DS_LIB=that .so): the product kernel carries neither.  Results of ablated builds are
wrong by construction -- only the times matter (profiles/r03_notes.md has them)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import _lib, ops
_lib.LIB_PATH = os.environ["DS_LIB"]
lib = _lib.load()
SH = ((56, 64, 192, 1), (56, 192, 64, 0), (28, 96, 128, 1))
for (hw, ci, co, stats) in SH:
    B = 256
    x = torch.relu(torch.randn(B, hw, hw, ci, device="cuda"))
    w = torch.randn(3, 3, ci, co, device="cuda") * 0.05
    z = torch.empty(B * hw * hw, co, device="cuda")
    p = ops.WinoPlan(B, hw, hw, ci, ci, co, co, flags=ops.DS_EPI_STATS if stats else 0, f4=True)
    u = torch.empty(p.u_elems, device="cuda")
    ops.wino_transform_weights(ops._p(w), u, ci, co, False, f4=True)
    st = torch.zeros(2 * co * max(p.partials, 1) + 16, device="cuda")
    pv = torch.zeros(co, device="cuda")
    f = lambda: p.run(ops._p(x), ops._p(u), ops._p(z), stats=ops._p(st), pivot=ops._p(pv))
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    lib.ds_debug_wino4_prof.argtypes = [C.c_void_p]
    lib.ds_debug_wino4_prof(out)
    t = [v * 0.01 for v in out]       # us (100 MHz)
    d = [t[i + 1] - t[i] for i in range(7)]
    print("%dx%d %3d->%3d launch %6.1f us | wg %5.1f: prologue %4.1f, loop %5.1f (%4.2f/step), epilogue %4.1f" % (
        hw, hw, ci, co, e0.elapsed_time(e1) * 100, t[7] - t[0], d[0], d[1], d[1] / (ci // 16), sum(d[2:])))
