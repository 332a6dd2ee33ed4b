import time, torch
x = torch.zeros(64, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(2000):
        x.add_(1.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue %.1f us/launch, total %.1f us/launch" % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
