import os, sys, time
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
net = SentimentNet(mode="joint", nb_emotions=15, rnn_size=512, vocab_size=10000, embedding_dim=300, post_size=32)
net.initialize(seed=1)
batch = to_device(synthetic_batch_numpy(256, 32, 10000, 15, seed=0))
main_prio = int(os.environ.get("DS_MAIN_PRIO", "0"))
ms = torch.cuda.Stream(priority=main_prio) if main_prio != 0 else torch.cuda.current_stream()
with torch.cuda.stream(ms):
    for _ in range(5): net.train_step(batch, 1e-3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): net.train_step(batch, 1e-3)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("DS_TEXT_PRIO=%s DS_MAIN_PRIO=%s: %.3f ms/step" % (os.environ.get("DS_TEXT_PRIO", "0"), main_prio, dt/20*1e3))
