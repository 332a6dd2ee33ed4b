#!/usr/bin/env python
"""pool_sums on / off (fp32 with pool_first and zcat off, or bf16): per-variable gradient differences of one training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
batch = to_device(synthetic_batch_numpy(B, 10, 50, seed=5))
res = []
for on in (True, False, False):
    net = SentimentNet(mode="joint", nb_emotions=15, rnn_size=32, vocab_size=50, embedding_dim=20, post_size=10, dtype=dtype)
    net.image.pool_sums = on
    if dtype == "f32":
        net.image.pool_first, net.image.zcat = False, False
    if len(res) == 2:
        net.image.bwd_sums = False           # every sum from a reduce pass: the scale of a pure summation-order change
    net.initialize(seed=7)
    net.train_step(batch, 1e-3)
    torch.cuda.synchronize()
    res.append(net.grads_state_dict())
rels = []
for name, g in res[1].items():
    r0 = np.linalg.norm(res[0][name].astype(np.float64) - g) / max(np.linalg.norm(g), 1e-30)
    r2 = np.linalg.norm(res[2][name].astype(np.float64) - g) / max(np.linalg.norm(g), 1e-30)
    rels.append((r0, r2, name))
for r0, r2, name in rels:
    if "beta" in name or "weights" in name:
        print("%-60s pool_sums %.2e   all-reduce-passes %.2e" % (name[-60:], r0, r2))
print("median %.2e worst %.2e | reference pair median %.2e worst %.2e" % (np.median([r[0] for r in rels]), max(r[0] for r in rels), np.median([r[1] for r in rels]), max(r[1] for r in rels)))
