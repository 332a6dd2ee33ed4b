import numpy as np, torch, sys
sys.path.insert(0, '.')
from oracle import tf_semantics as S
from tumblr_emotions_amd import ops
def dev(a): return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device='cuda')
for (N,H,W,Ci,Co,k) in [(2,14,14,24,64,3),(2,14,14,24,64,1),(2,14,14,32,64,3),(2,14,14,24,16,3),(8,14,14,24,64,3)]:
    rng = np.random.RandomState(3)
    w = rng.normal(size=(k,k,Ci,Co))*0.1
    dy = rng.normal(size=(N,H,W,Co))
    ref = S.conv2d_same_bwd_input(dy, w, (N,H,W,Ci), 1).reshape(-1,Ci)
    plan = ops.ConvPlan(N,H,W,Co,Co,k,k,1,Ci,Ci,Ci*Co,Co,1,flip=1)
    dx = torch.zeros(plan.M, Ci, device='cuda')
    plan.run(ops._p(dev(dy)), ops._p(dev(w)), ops._p(dx))
    torch.cuda.synchronize()
    err = np.abs(dx.cpu().numpy()-ref)
    bad_rows = np.where(err.max(1) > 1e-3)[0]
    bad_cols = np.where(err.max(0) > 1e-3)[0]
    print((N,H,W,Ci,Co,k), 'maxerr', err.max(), 'bad rows', len(bad_rows), bad_rows[:10], 'bad cols', bad_cols[:30])
    # per-tap check: which single-tap weights give right answers
    if k == 3:
        for tap in range(9):
            w1 = np.zeros_like(w); w1[tap//3, tap%3] = w[tap//3, tap%3]
            ref1 = S.conv2d_same_bwd_input(dy, w1, (N,H,W,Ci), 1).reshape(-1,Ci)
            dx.zero_()
            plan.run(ops._p(dev(dy)), ops._p(dev(w1)), ops._p(dx))
            torch.cuda.synchronize()
            print('   tap', tap, 'err', np.abs(dx.cpu().numpy()-ref1).max())
