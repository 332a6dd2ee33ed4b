import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
batch = to_device(synthetic_batch_numpy(16, 10, 50, seed=2))
res = {}
for sp in (True, False):
    for bs in (True, False):
        net = SentimentNet(mode="joint", nb_emotions=15, rnn_size=32, vocab_size=50, embedding_dim=20, post_size=10, dtype=dtype)
        net.image.stem_pool, net.image.bwd_sums = sp, bs
        net.initialize(seed=3)
        net.train_step(batch, 1e-3)
        torch.cuda.synchronize()
        pool = net.image.stages[1]
        res[(sp, bs)] = (pool.out.float().clone(), net.logits.detach().clone(), net.store.grad.clone(), net.image.stages[0].layer.pool_inside,
                         net.image.stages[0].layer.rstd.clone(), net.image.stages[0].layer.mean.clone())
        st = net.store
def cmp(a, b):
    ra, rb = res[a], res[b]
    rels = []
    for e in st.entries.values():
        if e.trainable:
            x, y = (g[e.offset:e.offset + e.numel].double() for g in (ra[2], rb[2]))
            rels.append(float((x - y).norm() / max(float(y.norm()), 1e-30)))
    print(a, "inside", ra[3], "vs", b, "inside", rb[3], ": pool.out max diff %.3e (neq %d of %d), logits %.3e, grads median %.2e worst %.2e, rstd %.2e mean %.2e"
          % (float((ra[0] - rb[0]).abs().max()), int((ra[0] != rb[0]).sum()), ra[0].numel(), float((ra[1] - rb[1]).abs().max()),
             float(np.median(rels)), max(rels), float((ra[4]-rb[4]).abs().max()), float((ra[5]-rb[5]).abs().max())))
cmp((True, True), (False, True))
cmp((True, True), (False, False))
cmp((False, True), (False, False))
cmp((True, False), (False, False))
