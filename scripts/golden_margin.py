#!/usr/bin/env python
"""How far the full-size joint step (BASELINE configs[2], tests/golden/joint_step_b256 fixture) sits from the fp64 oracle on the
gated gradients under LEGITIMATE reassociations of the HIP path: the statistics' partial grouping (stem with the pool inside or
not, Branch_3 fused or not), zcat, and the Winograd family (F(4x4) has 6.5x the rounding error of the direct fp32 conv).  Output:
one line per variant with max|dlogits|, |dloss| and the relative L2 of every gated gradient -- the measured margin behind the
gate in tests/test_golden_gpu.py (VERDICT r05 next #7c)."""
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch
import make_golden_fullsize as G
from tumblr_emotions_amd.net import SentimentNet

which = "joint"
g = np.load(os.path.join(ROOT, "tests", "golden", G.CFGS[which]["file"]))
cfg = json.loads(str(g["cfg"]))
params, emb, batch, mask = G.build(cfg)
names = [k[5:] for k in g.files if k.startswith("grad/")]
short = lambda n: "/".join(n.split("/")[-2:])
print("gated gradients: " + "; ".join(short(n) for n in names))
print("fixture spreads (oracle fp32 vs fp64): " + " ".join("%.2e" % float(g["spread/" + n]) for n in names))
variants = [dict(stem_pool=a, fuse_branch3=b, zcat=c, winograd4=True) for a, b, c in itertools.product((True, False), repeat=3)]
variants += [dict(stem_pool=True, fuse_branch3=True, zcat=True, winograd4=False), dict(stem_pool=False, fuse_branch3=False, zcat=True, winograd4=False),
             dict(stem_pool=True, fuse_branch3=True, zcat=True, winograd4=True, winograd=False)]
for v in variants:
    net = SentimentNet(mode=which, nb_emotions=15, im_features_size=256, rnn_size=cfg["H"], fc_size=512,
                       vocab_size=cfg["V"], embedding_dim=cfg["D"], post_size=cfg["T"])
    for k, val in v.items():
        setattr(net.image, k, val)
    sd = dict(params)
    sd["Text/W_embedding"] = emb
    net.load_state_dict(sd)
    dev = {k: torch.from_numpy(x).cuda() for k, x in batch.items()}
    net.train_step(dev, cfg["lr"], dropout_mask=torch.tensor(mask, dtype=torch.float32).cuda())
    torch.cuda.synchronize()
    dl = np.abs(net.logits.detach().cpu().numpy() - g["logits"]).max()
    dloss = abs(net.total_loss_value() - float(g["loss"]))
    grads = net.grads_state_dict()
    rels = []
    for n in names:
        ref = g["grad/" + n].astype(np.float64)
        got = grads[n].reshape(-1)
        got = got[::cfg["stride"]] if got.size > cfg["big"] else got
        rels.append(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))
    print(" ".join("%s=%d" % (k[:6], int(x)) for k, x in v.items()), "| dlogits %.2e dloss %.2e |" % (dl, dloss),
          " ".join("%.2e" % r for r in rels))
    del net
    torch.cuda.empty_cache()
