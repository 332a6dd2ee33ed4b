import sys
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd import ops
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for M, C in [(802816, 192), (802816, 64), (200704, 176), (200704, 128), (200704, 32), (200704, 288), (200704, 192), (200704, 96), (200704, 64), (50176, 304), (50176, 208), (50176, 48), (50176, 64), (50176, 296), (50176, 224), (50176, 448), (50176, 320), (50176, 128), (12544, 448), (12544, 320), (12544, 128), (12544, 624), (12544, 384)]:
    z = torch.randn(M, C, device='cuda'); dy = torch.randn(M, C, device='cuda'); y = torch.empty_like(z)
    mean = torch.zeros(C, device='cuda'); rstd = torch.ones(C, device='cuda'); shift = torch.zeros(C, device='cuda')
    coef = torch.zeros(2, C, device='cuda'); dbeta = torch.zeros(C, device='cuda')
    P = ops.bn_bwd_partials(M, C); part = torch.empty(2 * C * P, device='cuda')
    segs = ops.make_segments([(0, C, dy.data_ptr(), C)]); ysegs = ops.make_segments([(0, C, y.data_ptr(), C)])
    E = M * C * 4 / 1e9
    t1 = timeit(lambda: ops.bn_apply_relu(z, M, C, rstd, shift, ysegs))
    t2 = timeit(lambda: ops.bn_bwd_reduce(z, segs, M, C, mean, rstd, shift, part))
    t3 = timeit(lambda: ops.bn_bwd_apply(z, segs, M, C, mean, rstd, shift, coef, y))
    print("M=%7d C=%4d (%.0f MB): apply %.1f us %.2f TB/s | bwd_reduce %.1f us %.2f TB/s | bwd_apply %.1f us %.2f TB/s" % (M, C, E*1e3, t1*1e3, 2*E/t1, t2*1e3, 2*E/t2, t3*1e3, 3*E/t3))
