#!/usr/bin/env python
"""The pools of the joint step (B = 256) one by one: us per launch and TB/s of algorithmic bytes.  DS_LIB=<.so> for A/B."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import _lib, ops

if os.environ.get("DS_LIB"):
    _lib.LIB_PATH = os.environ["DS_LIB"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256


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


tot = {}
print("%-34s %9s %8s" % ("pool", "us", "TB/s"))
for hw, c in ((28, 192), (28, 256), (14, 480), (14, 512), (14, 512), (14, 512), (14, 528), (7, 832), (7, 832)):
    x = torch.randn(B, hw, hw, c, device="cuda")
    y = torch.empty_like(x)
    am = torch.empty(B, hw, hw, c, dtype=torch.uint8, device="cuda")
    dy = torch.randn_like(x)
    dx = torch.empty_like(x)
    n = x.numel()
    t = timeit(lambda: ops.maxpool_fwd(x, y, am, B, hw, hw, c, 3, 1, "SAME"))
    tb = timeit(lambda: ops.maxpool_bwd(dy, am, dx, False, B, hw, hw, c, 3, 1, "SAME"))
    tot["s1 fwd"] = tot.get("s1 fwd", 0) + t
    tot["s1 bwd"] = tot.get("s1 bwd", 0) + tb
    print("%-34s %9.1f %8.2f" % ("3x3/1 fwd  %dx%d C=%d" % (hw, hw, c), t, 9 * n / t / 1e6))
    print("%-34s %9.1f %8.2f" % ("3x3/1 bwd  %dx%d C=%d" % (hw, hw, c), tb, 9 * n / tb / 1e6))
for hw, c in ((112, 64), (56, 192), (28, 480)):
    z = torch.randn(B, hw, hw, c, device="cuda")
    oh = (hw + 1) // 2
    y = torch.empty(B, oh, oh, c, device="cuda")
    am = torch.empty(B, oh, oh, c, dtype=torch.uint8, device="cuda")
    rstd, shift = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda")
    t = timeit(lambda: ops.maxpool_bn_relu_fwd(z, rstd, shift, y, am, B, hw, hw, c, 3, 2))
    tot["s2 fwd"] = tot.get("s2 fwd", 0) + t
    print("%-34s %9.1f %8.2f" % ("3x3/2 bn+relu fwd %dx%d C=%d" % (hw, hw, c), t, (4 * z.numel() + 5 * y.numel()) / t / 1e6))
print("sums: " + ", ".join("%s %.1f us" % kv for kv in tot.items()))
