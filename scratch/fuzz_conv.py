"""Randomised parity sweep of ds_conv_igemm / ds_conv_wgrad against the NumPy oracle (fp64)."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from oracle import tf_semantics as S
from tumblr_emotions_amd import ops, _lib
lib = _lib.load()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
bad = 0
for case in range(ncases):
    k = int(rng.choice([1, 1, 3, 3, 5]))
    stride = int(rng.choice([1, 1, 1, 2]))
    N = int(rng.randint(1, 5)); H = int(rng.randint(max(k, 2), 20)); W = int(rng.randint(max(k, 2), 20))
    Ci = int(rng.choice([4, 8, 12, 16, 20, 24, 32, 36, 48, 64, 96, 100, 3, 7])); Co = int(rng.choice([4, 8, 15, 16, 32, 40, 64, 96, 100, 160, 200]))
    path = int(rng.choice([0, 0, 1, 3])); nt = int(rng.choice([0, 0, 1, 2, 3]))
    x = rng.normal(size=(N, H, W, Ci)); w = rng.normal(size=(k, k, Ci, Co)) * 0.2
    ref = S.conv2d_same(x, w, stride)
    OH, OW = ref.shape[1], ref.shape[2]
    lib.ds_conv_set_path(path); lib.ds_conv_set_tile(1 if nt else 0, nt)
    flags = int(rng.choice([0, ops.DS_EPI_STATS, ops.DS_EPI_BIAS | ops.DS_EPI_RELU, ops.DS_EPI_ACCUM]))
    bias = rng.normal(size=Co); prev = rng.normal(size=(N * OH * OW, Co))
    plan = ops.ConvPlan(N, H, W, Ci, Ci, k, k, stride, Co, Co, Ci * Co, 1, Co, flags=flags)
    z = dev(prev) if flags & ops.DS_EPI_ACCUM else torch.empty(plan.M, Co, device="cuda")
    stats = torch.zeros(2, Co, max(plan.partials, 1), device="cuda")
    xd, wd, bd = dev(x), dev(w), dev(bias)
    plan.run(ops._p(xd), ops._p(wd), ops._p(z), bias=ops._p(bd), stats=ops._p(stats))
    want = ref.reshape(-1, Co)
    if flags & ops.DS_EPI_BIAS: want = np.maximum(want + bias, 0)
    if flags & ops.DS_EPI_ACCUM: want = want + prev
    torch.cuda.synchronize()
    got = z.cpu().numpy().astype(np.float64)
    tol = 3e-4 * max(1.0, np.abs(want).max())
    ok = np.abs(got - want).max() <= tol
    if flags & ops.DS_EPI_STATS:
        ok = ok and np.abs(stats[0].sum(1).cpu().numpy() - want.sum(0)).max() <= 2e-3 * max(1.0, np.abs(want.sum(0)).max())
        ok = ok and np.abs(stats[1].sum(1).cpu().numpy() - (want ** 2).sum(0)).max() <= 2e-3 * max(1.0, (want ** 2).sum(0).max())
    # dgrad (stride 1 only) and wgrad of the same geometry
    if stride == 1:
        dy = rng.normal(size=(N, OH, OW, Co))
        g = ops.ConvPlan(N, H, W, Co, Co, k, k, 1, Ci, Ci, Ci * Co, Co, 1, flip=1)
        dx = torch.empty(g.M, Ci, device="cuda"); dyd = dev(dy)
        g.run(ops._p(dyd), ops._p(wd), ops._p(dx))
        dref = S.conv2d_same_bwd_input(dy, w, (N, H, W, Ci), 1).reshape(-1, Ci)
        torch.cuda.synchronize()
        ok = ok and np.abs(dx.cpu().numpy() - dref).max() <= 3e-4 * max(1.0, np.abs(dref).max())
        wp = ops.WgradPlan(N, H, W, Ci, Ci, k, k, 1, Co, Co)
        ws = torch.empty(max(wp.ws_bytes // 4, 1), device="cuda"); dw = torch.empty(k, k, Ci, Co, device="cuda")
        wp.run(ops._p(xd), ops._p(dyd), ops._p(dw), ops._p(ws), wp.ws_bytes)
        wref = S.conv2d_same_bwd_filter(x, dy, (k, k, Ci, Co), 1)
        torch.cuda.synchronize()
        ok = ok and np.abs(dw.cpu().numpy() - wref).max() <= 5e-4 * max(1.0, np.abs(wref).max())
    if not ok:
        bad += 1
        print("MISMATCH case %d: N=%d H=%d W=%d Ci=%d Co=%d k=%d stride=%d path=%d nt=%d flags=%d" % (case, N, H, W, Ci, Co, k, stride, path, nt, flags), flush=True)
lib.ds_conv_set_path(0); lib.ds_conv_set_tile(0, 0)
print("fuzz: %d cases, %d mismatches" % (ncases, bad))
