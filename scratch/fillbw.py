import sys; sys.path.insert(0, ".")
import torch, bench, json
from tumblr_emotions_amd import ops
n = 8192 * 128 * 300
out = torch.empty(n, device="cuda")
src = torch.randn(n, device="cuda")
def t(f, reps=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
us = t(lambda: out.fill_(1.0)); print("torch fill   %.1f us  %.0f GB/s written" % (us, n * 4 / us / 1e3))
us = t(lambda: ops.fill(out, n, 1.0)); print("ds_fill      %.1f us  %.0f GB/s written" % (us, n * 4 / us / 1e3))
us = t(lambda: out.copy_(src)); print("torch copy   %.1f us  %.0f GB/s written (+ same read)" % (us, n * 4 / us / 1e3))
print(json.dumps(bench.gather_bandwidth()))
