#!/usr/bin/env python
"""ds_maxpool_bwd (accumulating) against ds_maxpool3_bwd_sums on the Branch_3 pools of the 16-bit step (B = 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(28, 256, "bf16"), (14, 480, "bf16"), (14, 512, "bf16"), (14, 528, "bf16"), (7, 832, "f32"), (7, 832, "bf16")]


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


for hw, c, dt in SHAPES:
    x = torch.randn(B, hw, hw, c, device="cuda")
    pooled, am = torch.empty_like(x), torch.empty(B, hw, hw, c, dtype=torch.uint8, device="cuda")
    ops.maxpool_fwd(x, pooled, am, B, hw, hw, c, 3, 1, "SAME")
    dy, dx = torch.randn_like(x), torch.randn_like(x)
    y = torch.relu(torch.randn_like(x))
    if dt == "bf16":
        y = y.to(torch.bfloat16)
    P = ops.maxpool3_bwd_sums_partials(B, hw, c)
    part = torch.empty(2 * c * P, device="cuda")
    t0 = timeit(lambda: ops.maxpool_bwd(dy, am, dx, True, B, hw, hw, c, 3, 1, "SAME"))
    t1 = timeit(lambda: ops.maxpool3_bwd_sums(dy, am, dx, True, y, B, hw, hw, c, part))
    n = B * hw * hw * c
    print("%2dx%-2d C=%4d y=%-4s P=%5d | rolling %7.1f us (%.2f TB/s)   with sums %7.1f us (%.2f TB/s)"
          % (hw, hw, c, dt, P, t0, n * 13 / t0 / 1e6, t1, n * (15 if dt == "bf16" else 17) / t1 / 1e6))
