import sys, time
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd import ops
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
x = torch.zeros(1024, device='cuda')
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5000): ops.fill(x, 1024, 1.0)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host cost of one ops.fill launch: %.2f us" % ((t1 - t0) / 5000 * 1e6))
t0 = time.perf_counter()
for _ in range(20000): ops._stream()
print("_stream(): %.2f us" % ((time.perf_counter() - t0) / 20000 * 1e6))
for B in (32, 256):
    net = SentimentNet(mode="joint", nb_emotions=15, rnn_size=512, vocab_size=10000, embedding_dim=300, post_size=32)
    net.initialize(seed=1)
    batch = to_device(synthetic_batch_numpy(B, 32, 10000, 15, seed=0))
    for _ in range(5): net.train_step(batch, 1e-3)
    torch.cuda.synchronize()
    # host enqueue time: time the python side only, with the GPU far behind (queue is deep enough for one step)
    t0 = time.perf_counter(); net.train_step(batch, 1e-3); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    n = 20; t3 = time.perf_counter()
    for _ in range(n): net.train_step(batch, 1e-3)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print("B=%d: host enqueue of one step %.2f ms, step time %.2f ms" % (B, (t1 - t0) * 1e3, (t4 - t3) / n * 1e3))
    del net
