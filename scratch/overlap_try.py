"""Can an HBM-bound kernel make progress while the persistent conv kernel owns the CUs?"""
import sys, ctypes as C
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd import ops, _lib
lib = _lib.load()
N,H,W,Ci,Co,k = 256,56,56,192,64,3
plan = ops.ConvPlan(N,H,W,Ci,Ci,k,k,1,Co,Co,Ci*Co,Ci,1,flip=1)      # conv2c dgrad (k-contig)
x = torch.randn(N*H*W, Ci, device='cuda'); w = torch.randn(k*k*Ci*Co, device='cuda')*0.05; z = torch.empty(N*H*W, Co, device='cuda')
M2, C2 = 802816, 192
zb = torch.randn(M2, C2, device='cuda'); yb = torch.empty_like(zb)
rstd = torch.ones(C2, device='cuda'); shift = torch.zeros(C2, device='cuda')
segs = ops.make_segments([(0, C2, yb.data_ptr(), C2)])
def conv(): plan.run(ops._p(x), ops._p(w), ops._p(z))
def bn(): ops.bn_apply_relu(zb, M2, C2, rstd, shift, segs)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); 
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
s2 = torch.cuda.Stream()
def both(nbn):
    main = torch.cuda.current_stream()
    s2.wait_stream(main)
    with torch.cuda.stream(s2):
        for _ in range(nbn): bn()
    conv()
    main.wait_stream(s2)
for mt, nt in [(0,0),(1,1),(2,2)]:
    lib.ds_conv_set_tile(mt, nt)
    tc, tb = timeit(conv), timeit(bn)
    print("tile %d,%d: conv %.3f ms  bn %.3f ms | conv + 4 bn serial %.3f | concurrent %.3f" % (mt, nt, tc, tb, tc + 4*tb, timeit(lambda: both(4))))
