import sys; sys.path.insert(0, ".")
import torch
from tumblr_emotions_amd import ops
B, hw = 16, 14
def timeit(f, reps=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for ci, co in [(16, 32), (32, 32), (64, 32), (144, 32), (288, 32), (144, 288)]:
    x = torch.randn(B, hw, hw, ci, device="cuda"); w = torch.randn(3, 3, ci, co, device="cuda") * 0.05
    z = torch.empty(B * hw * hw, co, device="cuda"); u = torch.empty(16, co, ci, device="cuda")
    ops.wino_transform_weights(ops._p(w), u, ci, co, False)
    row = []
    for fl in (0, 2048, 256, 512, 256 | 512, 256 | 512 | 1024):
        p = ops.WinoPlan(B, hw, hw, ci, ci, co, co, flags=fl)
        row.append(timeit(lambda: p.run(ops._p(x), ops._p(u), ops._p(z))))
    print("Cin %4d Cout %4d ksteps %3d | full %6.1f  noshare %6.1f  noAload %6.1f  noBdma %6.1f  noloads %6.1f  nothing %6.1f us" % ((ci, co, ci // 8) + tuple(row)))
