#!/usr/bin/env python
"""What the BatchNorm-sums epilogue costs the F(4x4) Winograd dgrads: every 3x3 dgrad shape of the joint step (B = 256)
through ds_conv_wino4 with flags 0 and with DS_EPI_BNSUMS.  us per launch.  DS_LIB=<.so> for A/B of kernel variants."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import _lib, ops

if os.environ.get("DS_LIB"):
    _lib.LIB_PATH = os.environ["DS_LIB"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
# (map, layer Cin, layer Cout): dgrad runs Cout -> Cin
SHAPES = [(56, 64, 192), (28, 96, 128), (28, 16, 32), (28, 128, 192), (28, 32, 96), (14, 96, 208), (14, 112, 224), (14, 128, 256),
          (14, 144, 288), (14, 160, 320), (14, 32, 64), (14, 32, 128), (7, 160, 320), (7, 32, 128)]


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


tot = [0.0, 0.0]
print("%4s %5s %5s | %9s %9s" % ("HW", "K", "N", "plain", "bnsums"))
for hw, ci, co in SHAPES:
    if co % 16 or ci % 4:
        continue
    M = B * hw * hw
    dz = torch.randn(M, co, device="cuda")
    w = torch.randn(3, 3, ci, co, device="cuda") * 0.05
    dx = torch.empty(M, ci, device="cuda")
    y = torch.relu(torch.randn(M, ci, device="cuda"))
    res = []
    for bns in (False, True):
        plan = ops.WinoPlan(B, hw, hw, co, co, ci, ci, flags=0, f4=True)
        P = plan.enable_bnsums() if bns else 0
        u = torch.empty(plan.u_elems, device="cuda")
        ops.wino_transform_weights(ops._p(w), u, ci, co, True, f4=True)
        sums = torch.zeros(2 * ci * max(P, 1) + 16, device="cuda")
        res.append(timeit(lambda: plan.run(ops._p(dz), ops._p(u), ops._p(dx), stats=ops._p(sums) if bns else None, ymask=ops._p(y) if bns else None)))
    tot[0] += res[0]; tot[1] += res[1]
    print("%4d %5d %5d | %9.1f %9.1f" % (hw, co, ci, res[0], res[1]))
print("sum: plain %.1f us, with sums %.1f us" % tuple(tot))
