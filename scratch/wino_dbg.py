import sys; sys.path.insert(0, ".")
import numpy as np, torch
from oracle import tf_semantics as S
from tumblr_emotions_amd import ops
N, H, W, Ci, Co = 1, 8, 8, 8, 32
rng = np.random.RandomState(0)
x = rng.normal(size=(N, H, W, Ci)); w = rng.normal(size=(3, 3, Ci, Co)) * 0.1
ref = S.conv2d_same(x, w, 1)
xd = torch.tensor(x, dtype=torch.float32, device="cuda"); wd = torch.tensor(w, dtype=torch.float32, device="cuda")
u = torch.empty(16, Co, Ci, device="cuda"); ops.wino_transform_weights(ops._p(wd), u, Ci, Co, False)
for fl in (2048, 0):
    z = torch.zeros(N * H * W, Co, device="cuda")
    ops.WinoPlan(N, H, W, Ci, Ci, Co, Co, flags=fl).run(ops._p(xd), ops._p(u), ops._p(z))
    torch.cuda.synchronize()
    err = np.abs(z.cpu().numpy().reshape(H, W, Co) - ref[0]).max(axis=2)
    print("flags", fl); print(np.round(err, 3))
