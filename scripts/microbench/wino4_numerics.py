# F(4x4,3x3) vs F(2x2,3x3) in fp32 against an fp64 direct correlation: error on activations-like data
import numpy as np
BT = np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]], np.float64)
G = np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]], np.float64)
AT = np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]], np.float64)
BT2 = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], np.float64)
G2 = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], np.float64)
AT2 = np.array([[1,1,1,0],[0,1,-1,-1]], np.float64)
rng = np.random.RandomState(0)
Cin, Cout, T = 192, 64, 200
for name, (bt, g, at, m) in {"F2": (BT2, G2, AT2, 2), "F4": (BT, G, AT, 4)}.items():
    a = m + 2
    d = np.maximum(rng.standard_normal((T, Cin, a, a)), 0).astype(np.float32)     # post-ReLU activations
    w = (rng.standard_normal((Cin, Cout, 3, 3)) * 0.05).astype(np.float32)
    ref = np.zeros((T, Cout, m, m))
    for i in range(m):
        for j in range(m):
            ref[:, :, i, j] = np.einsum("tcxy,coxy->to", d[:, :, i:i+3, j:j+3].astype(np.float64), w.astype(np.float64))
    f = np.float32
    U = np.einsum("ax,coxy,by->coab", g.astype(f), w, g.astype(f)).astype(f)
    V = np.einsum("ax,tcxy,by->tcab", bt.astype(f), d, bt.astype(f)).astype(f)
    M = np.einsum("tcab,coab->toab", V, U).astype(f)      # fp32 accumulate (numpy pairwise-ish)
    Y = np.einsum("ia,toab,jb->toij", at.astype(f), M, at.astype(f)).astype(f)
    err = np.abs(Y - ref)
    print(name, "max|ref|", np.abs(ref).max(), "max err", err.max(), "rms err", np.sqrt((err**2).mean()), "rel rms", np.sqrt((err**2).mean()) / np.sqrt((ref**2).mean()))
# direct fp32 for scale
acc = np.zeros((T, Cout, 4, 4), np.float32)
for i in range(4):
    for j in range(4):
        acc[:, :, i, j] = np.einsum("tcxy,coxy->to", d[:, :, i:i+3, j:j+3], w)
print("direct fp32 max err", np.abs(acc - ref).max(), "rel rms", np.sqrt(((acc-ref)**2).mean()) / np.sqrt((ref**2).mean()))
