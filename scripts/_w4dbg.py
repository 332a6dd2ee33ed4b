import os, sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from tumblr_emotions_amd import _lib, ops
B, hw, ci, co = int(sys.argv[1]), 28, 96, 128
torch.manual_seed(1)
x = torch.relu(torch.randn(B, hw, hw, ci, device="cuda"))
w = torch.randn(3, 3, ci, co, device="cuda") * 0.05
pivot = torch.randn(co, device="cuda") * 0.1
zs = {}
for flags in (0, ops.DS_EPI_STATS):
    p = ops.WinoPlan(B, hw, hw, ci, ci, co, co, flags=flags, f4=True)
    u = torch.empty(p.u_elems, device="cuda")
    ops.wino_transform_weights(ops._p(w), u, ci, co, False, f4=True)
    st = torch.zeros(2 * co * max(p.partials, 1) + 16, device="cuda")
    z = torch.zeros(B * hw * hw, co, device="cuda")
    p.run(ops._p(x), ops._p(u), ops._p(z), stats=ops._p(st), pivot=ops._p(pivot))
    torch.cuda.synchronize()
    zs[flags] = z.cpu().numpy().reshape(B, hw, hw, co)
d = np.abs(zs[0] - zs[ops.DS_EPI_STATS])
print("max diff", d.max(), "count", (d > 1e-4).sum(), "of", d.size)
idx = np.argwhere(d > 1e-4)
print("channels", np.unique(idx[:, 3])[:40])
print("rows", np.unique(idx[:, 1])[:40], "cols", np.unique(idx[:, 2])[:40], "imgs", np.unique(idx[:, 0]))
print(idx[:10])
