import sys, time
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
net = SentimentNet(mode="joint", nb_emotions=15, rnn_size=512, vocab_size=10000, embedding_dim=300, post_size=32)
net.initialize(seed=1)
batch = to_device(synthetic_batch_numpy(256, 32, 10000, 15, seed=0))
for _ in range(3): net.train_step(batch, 1e-3)
torch.cuda.synchronize()
def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager ms/step", timeit(lambda: net.train_step(batch, 1e-3)))
# host-only cost: how long does it take to enqueue one step?
t0 = time.perf_counter(); net.train_step(batch, 1e-3); t1 = time.perf_counter(); torch.cuda.synchronize()
print("host enqueue ms", (t1 - t0) * 1e3)
