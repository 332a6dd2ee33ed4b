"""How much would batching the three branch BatchNorm launches of a Mixed block into one save?"""
import sys
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd import ops
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for M, Cs in [(200704, (128, 32, 32)), (50176, (208, 48, 64)), (50176, (320, 128, 128)), (12544, (384, 128, 128))]:
    ts = {}
    bufs = []
    for C in list(Cs) + [sum(Cs)]:
        z = torch.randn(M, C, device='cuda'); dy = torch.randn(M, C, device='cuda'); y = torch.empty_like(z)
        mean = torch.zeros(C, device='cuda'); rstd = torch.ones(C, device='cuda'); shift = torch.zeros(C, device='cuda')
        coef = torch.zeros(2, C, device='cuda'); dbeta = torch.zeros(C, device='cuda'); beta = torch.zeros(C, device='cuda')
        P = ops.bn_bwd_partials(M, C); part = torch.empty(2 * C * P, device='cuda'); stats = torch.zeros(2 * C * 392, device='cuda')
        segs = ops.make_segments([(0, C, dy.data_ptr(), C)]); ysegs = ops.make_segments([(0, C, y.data_ptr(), C)])
        bufs.append((z, dy, y, mean, rstd, shift, coef, dbeta, beta, P, part, stats, segs, ysegs, C))
    def fwd(b):
        z, dy, y, mean, rstd, shift, coef, dbeta, beta, P, part, stats, segs, ysegs, C = b
        ops.bn_finalize(stats, 392, M, C, beta, 1e-3, 0.9997, mean, rstd, shift, None, None)
        ops.bn_apply_relu(z, M, C, rstd, shift, ysegs)
    def bwd(b):
        z, dy, y, mean, rstd, shift, coef, dbeta, beta, P, part, stats, segs, ysegs, C = b
        ops.bn_bwd_reduce(z, segs, M, C, mean, rstd, shift, part)
        ops.bn_bwd_finalize(part, P, M, C, dbeta, coef)
        ops.bn_bwd_apply(z, segs, M, C, mean, rstd, shift, coef, y)
    f3 = timeit(lambda: [fwd(b) for b in bufs[:3]]); f1 = timeit(lambda: fwd(bufs[3]))
    b3 = timeit(lambda: [bwd(b) for b in bufs[:3]]); b1 = timeit(lambda: bwd(bufs[3]))
    print("M=%6d C=%s: fwd 3 layers %.1f us vs one of the summed width %.1f us | bwd %.1f vs %.1f us" % (M, Cs, f3, f1, b3, b1))
