import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import ops, _lib
_lib.load().ds_debug_conv_wino_allow_ablation(1)
B = 256
def timeit(f, reps=10):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (hw, ci, co) in [(56, 64, 192), (56, 192, 64), (28, 128, 192), (14, 160, 320), (14, 320, 160)]:
    x = torch.randn(B, hw, hw, ci, device="cuda")
    u = torch.randn(16, co, ci, device="cuda")
    z = torch.empty(B * hw * hw, co, device="cuda")
    out = []
    for fl in (0, 256, 512, 768, 1024, 1024 + 768):
        p = ops.WinoPlan(B, hw, hw, ci, ci, co, co, flags=fl)
        out.append(timeit(lambda: p.run(ops._p(x), ops._p(u), ops._p(z))))
    items = ((B * ((hw + 1) // 2) ** 2 + 127) // 128) * ((co + 31) // 32)
    rounds = items / 256.0
    mf = (ci // 8) * 64 * 64 / 2.4e3   # us of MFMA per item at 2.4 GHz
    print(hw, ci, co, "items %d rounds %.2f mfma/item %.1f us |" % (items, rounds, mf), " ".join("%8.1f" % t for t in out), "| per item", " ".join("%6.1f" % (t / rounds) for t in out))
