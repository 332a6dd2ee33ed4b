#!/usr/bin/env python
"""How far is a fp32 evaluation of the Deep Sentiment step from the fp64 one -- for the ORACLE against itself?

    python scripts/oracle_fp32_spread.py 4 16 64        # batch sizes

Finding (profiles/r02_oracle_fp32_spread.txt): logits and loss agree to ~1e-5, but the gradients of everything
below a ReLU / max-pool differ by ~1e-2 relative L2 at EVERY batch size.  Mechanism: deep in the tower the fp32
forward activations carry ~3e-5 relative error, so a ~1e-5 fraction f of the ReLU decisions of a layer flips
(pre-activation within rounding of zero); each flip removes / adds one element of the gradient, which moves the
gradient of that layer's input by ~sqrt(f) in relative L2 -- independent of the number of elements.  The
per-layer table of scripts/oracle_fp32_spread_layers.py shows the jump happening across ONE conv+BN+ReLU backward (1.7e-6 above
Mixed_5c's last convs, 6e-3..1e-2 right below).  Consequence for the parity tests: a plain fp32-vs-fp64
comparison of tower gradients cannot be gated below a few percent, so tests/test_model_gpu.py evaluates the fp64
oracle along the decisions the HIP forward pass took (DeepSentimentRef.inject), where 1e-3 holds.
"""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tf_semantics as S
from oracle import torch_ref as R
for B in [int(a) for a in sys.argv[1:]] or [16]:
    print('== batch', B)
    rng = np.random.RandomState(7)
    V, D, H, T = 60, 20, 32, 12
    params = R.make_params("joint", rng, num_classes=15, im_features_size=256, embed_dim=D, rnn_size=H, fc_size=512, dtype=np.float64)
    for k in params:
        if k.endswith("beta"):
            params[k] = rng.normal(0, 0.1, size=params[k].shape)
    emb = S.synthetic_embedding(V, D).astype(np.float64)
    batch = S.synthetic_batch(B, T, V, seed=11)
    mask = (rng.uniform(size=(B, 1024)) < 0.8).astype(np.float64)
    ref = R.DeepSentimentRef(params, emb, "joint", torch.float64)
    ref32 = R.DeepSentimentRef(params, emb, "joint", torch.float32)
    t0 = time.time()
    o = ref.train_step(batch, 1e-3, torch.tensor(mask))
    t1 = time.time()
    o32 = ref32.train_step(batch, 1e-3, torch.tensor(mask, dtype=torch.float32))
    t2 = time.time()
    print("fp64 %.1fs fp32 %.1fs" % (t1 - t0, t2 - t1))
    print("logit diff", float((o32["logits"].double() - o["logits"]).abs().max()), "loss diff", abs(o32["loss"] - o["loss"]))
    rows = []
    for n, g in o["grads"].items():
        d = o32["grads"][n].double() - g
        rows.append((float(d.norm() / max(float(g.norm()), 1e-30)), float(d.abs().max() / max(float(g.abs().max()), 1e-30)), n))
    rows.sort(reverse=True)
    for r in rows[:12]:
        print("%.3e %.3e %s" % r)
    print("median relL2 %.3e" % np.median([r[0] for r in rows]))
    print("n>1e-3:", sum(r[0] > 1e-3 for r in rows), "of", len(rows))
