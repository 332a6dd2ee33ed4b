#!/usr/bin/env python
"""ds_conv_wgrad per layer (B = 256): the trainable layers of the reference freeze (Mixed_5c, the LSTM / FC
matrices) and a sample of the lower layers that train_all adds.  DS_WGRAD_DIRECT=0 selects the round-1 LDS kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
# (N, HW, Cin, Cout, k)
SHAPES = [(B, 7, 832, 624, 1), (B, 7, 192, 384, 3), (B, 7, 48, 128, 3), (B, 7, 832, 128, 1),
          (B * 32, 1, 300, 2048, 1), (B * 32, 1, 512, 2048, 1), (B, 1, 1024, 15, 1),
          (B, 56, 64, 192, 3), (B, 28, 96, 128, 3), (B, 28, 192, 176, 1), (B, 14, 480, 304, 1), (B, 14, 112, 224, 3),
          (B, 14, 160, 320, 3), (B, 7, 160, 320, 3)]


if os.environ.get("WGRAD_ONLY"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["WGRAD_ONLY"].split(",")]


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


tot = 0.0
for (n, hw, ci, co, k) in SHAPES:
    x = torch.randn(n * hw * hw, ci, device="cuda")
    dz = torch.randn(n * hw * hw, co, device="cuda")
    dw = torch.empty(k, k, ci, co, device="cuda")
    plan = ops.WgradPlan(n, hw, hw, ci, ci, k, k, 1, co, co)
    ws_bytes = max(plan.ws_bytes, 64 * k * k * ci * co * 4) if os.environ.get("DS_WGRAD_FORCE") else plan.ws_bytes
    ws = torch.empty(max(ws_bytes // 4, 1), device="cuda")
    t = timeit(lambda: plan.run(ops._p(x), ops._p(dz), ops._p(dw), ops._p(ws), ws_bytes))
    fl = 2.0 * n * hw * hw * k * k * ci * co
    tot += t
    print("%6d %3d %5d %5d %2d | %9.1f us %7.1f TF  ws %6.1f MB" % (n, hw, ci, co, k, t, fl / t / 1e6, plan.ws_bytes / 1e6))
print("sum %.1f us" % tot)
