#!/usr/bin/env python
"""Per-layer view of scripts/oracle_fp32_spread.py: relative L2 distance between the fp32 and the fp64 oracle of every
conv+BN+ReLU output (act) and of the gradient w.r.t. it (grad), image tower, B given on the command line."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tf_semantics as S
from oracle import torch_ref as R
B = int(sys.argv[1])
rng = np.random.RandomState(7)
params = R.make_params("image", rng, num_classes=15, dtype=np.float64)
batch = S.synthetic_batch(B, 8, 10, seed=11)
refs = {}
acts = {}
for dt in (torch.float64, torch.float32):
    ref = R.DeepSentimentRef(params, None, "image", dt)
    store = []
    orig = ref._cbr
    def cbr(x, scope, stride=1, orig=orig, store=store):
        y = orig(x, scope, stride)
        y.retain_grad()
        store.append((scope, y))
        return y
    ref._cbr = cbr
    logits = ref.forward(batch, None)
    total, ce = ref.loss(logits, batch["labels"])
    total.backward()
    acts[dt] = store
for (s, y64), (_, y32) in zip(acts[torch.float64], acts[torch.float32]):
    fa = float((y32.double() - y64).norm() / y64.norm())
    g64, g32 = y64.grad, y32.grad
    fg = float((g32.double() - g64).norm() / g64.norm())
    print("%-50s act relL2 %.2e  grad relL2 %.2e  |act| %.2e" % (s[12:], fa, fg, float(y64.abs().mean())))
