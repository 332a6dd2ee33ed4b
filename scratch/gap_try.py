import sys, time
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd import ops
x = torch.zeros(1024, device='cuda')
big = torch.zeros(64*1024*1024, device='cuda')
def chain(n, t):
    for _ in range(n): ops.fill(t, t.numel(), 1.0)
for name, t in (("4 KB fill", x), ("256 MB fill", big)):
    chain(50, t); torch.cuda.synchronize()
    n = 2000 if t is x else 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); chain(n, t); e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print("%s: %.2f us per launch on the GPU timeline, host enqueue %.2f us" % (name, e0.elapsed_time(e1)/n*1e3, (t1-t0)/n*1e6))
