"""HBM-bound kernel on a second stream under a one-tile-per-workgroup conv launch (28x28 map)."""
import sys, ctypes as C
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd import ops, _lib
lib = _lib.load()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
s2 = torch.cuda.Stream()
for (N,H,Ci,Co,k,kc,name) in [(256,28,192,128,3,True,"3b 3x3 dgrad"),(256,28,128,192,3,False,"3c 3x3 fwd"),(256,14,512,296,1,True,"4d fused dgrad"),(256,56,192,64,3,True,"conv2c dgrad (persistent)")]:
    if kc: plan = ops.ConvPlan(N,H,H,Ci,Ci,k,k,1,Co,Co,Ci*Co,Ci,1,flip=1)
    else:  plan = ops.ConvPlan(N,H,H,Ci,Ci,k,k,1,Co,Co,Ci*Co,1,Co)
    M = N*H*H
    x = torch.randn(M, Ci, device='cuda'); w = torch.randn(k*k*Ci*Co, device='cuda')*0.05; z = torch.empty(M, Co, device='cuda')
    C2 = 256
    zb = torch.randn(M, C2, device='cuda'); yb = torch.empty_like(zb)
    rstd = torch.ones(C2, device='cuda'); shift = torch.zeros(C2, device='cuda')
    segs = ops.make_segments([(0, C2, yb.data_ptr(), C2)])
    def conv(): plan.run(ops._p(x), ops._p(w), ops._p(z))
    def bn(): ops.bn_apply_relu(zb, M, C2, rstd, shift, segs)
    tc, tb = timeit(conv), timeit(bn)
    nb = max(1, int(round(0.5*tc/tb)))      # ~half the conv's duration of HBM-bound work
    def both():
        main = torch.cuda.current_stream()
        s2.wait_stream(main)
        with torch.cuda.stream(s2):
            for _ in range(nb): bn()
        conv()
        main.wait_stream(s2)
    tt = timeit(both)
    print("%-28s conv %.3f ms, bn %.3f ms x%d | serial %.3f | concurrent %.3f  (hidden %.0f%% of the bn time)" % (name, tc, tb, nb, tc+nb*tb, tt, 100*(tc+nb*tb-tt)/(nb*tb)))
