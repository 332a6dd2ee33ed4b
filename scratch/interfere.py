"""How much does a chain of small GEMMs on a second stream slow the persistent conv kernel?"""
import sys, ctypes as C
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd import ops, _lib
lib = _lib.load()
N,H,W,Ci,Co,k = 256,56,56,64,192,3
plan = ops.ConvPlan(N,H,W,Ci,Ci,k,k,1,Co,Co,Ci*Co,1,Co,flags=ops.DS_EPI_STATS)      # conv2c fwd
x = torch.randn(N*H*W, Ci, device='cuda'); w = torch.randn(k*k*Ci*Co, device='cuda')*0.05; z = torch.empty(N*H*W, Co, device='cuda')
stats = torch.zeros(2, Co, plan.partials, device='cuda')
def conv(): plan.run(ops._p(x), ops._p(w), ops._p(z), stats=ops._p(stats))
B, Hh = 256, 512
g = ops.gemm_plan(B, Hh, 4*Hh, Hh, 4*Hh, 4*Hh, False, splits=4, z_split_stride=B*4*Hh)
h = torch.randn(B, Hh, device='cuda'); wh = torch.randn(Hh, 4*Hh, device='cuda')*0.02; slabs = torch.empty(4, B, 4*Hh, device='cuda')
tmp = torch.empty(B, 4*Hh, device='cuda')
def chain(n=31):
    for _ in range(n):
        g.run(ops._p(h), ops._p(wh), ops._p(slabs))
        ops.fill(tmp, tmp.numel(), 0.0)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
s2 = torch.cuda.Stream()
conv_ms = []
def both():
    main = torch.cuda.current_stream()
    s2.wait_stream(main)
    with torch.cuda.stream(s2):
        chain()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); conv(); b.record()
    conv_ms.append((a, b))
    main.wait_stream(s2)
for sched, mt, nt in [(0,0,0),(1,0,0),(2,0,0),(0,1,1),(1,1,1),(2,1,1)]:
    lib.ds_conv_set_tile(mt, nt); lib.ds_conv_set_sched(sched)
    tc, tb = timeit(conv), timeit(chain)
    conv_ms.clear()
    tt = timeit(both)
    torch.cuda.synchronize()
    cm = sum(a.elapsed_time(b) for a, b in conv_ms[1:]) / (len(conv_ms) - 1)
    print("sched %d tile %d,%d: conv %.3f ms  chain %.3f ms | serial %.3f | concurrent total %.3f, conv inside %.3f" % (sched, mt, nt, tc, tb, tc + tb, tt, cm))
