import numpy as np, torch, sys
sys.path.insert(0, '.')
from tumblr_emotions_amd import ops
def dev(a): return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device='cuda')
rng = np.random.RandomState(0)
for (M,K,N) in [(3,15,1024),(3,16,1024),(70,15,200),(3,15,128),(3,15,64)]:
    dy = rng.normal(size=(M,K)); w = rng.normal(size=(N,K))
    dyd, wd = dev(dy), dev(w)
    out = torch.zeros(M,N,device='cuda')
    ops.gemm_plan(M,K,N,K,N,K,transposed_w=True).run(ops._p(dyd), ops._p(wd), ops._p(out))
    torch.cuda.synchronize()
    e = np.abs(out.cpu().numpy() - dy@w.T)
    print((M,K,N), 'maxerr', e.max(), 'bad cols', np.where(e.max(0)>1e-3)[0][:20], 'n bad', (e.max(0)>1e-3).sum())
