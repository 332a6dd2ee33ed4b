#!/usr/bin/env python
"""Adds to tests/golden/joint_step_b256_oracle.npz what a PLAIN fp32-vs-fp64 comparison of the full-size joint step needs to be
exact about the dense layer's ReLU (relu(concat W_fc + b_fc), im_text_rnn_model.py:98-101; 256 x 512 units at B = 256):

  dense_mask           the fp64 oracle's decisions (pre-activation > 0), bit-packed
  flip/units           [K, 2] (sample, unit) of the units whose pre-activation lies within 1e-4 of zero -- decisions an fp32
                       evaluation cannot make (the HIP path's forward noise on these pre-activations is 4e-5,
                       profiles/r06_notes.md; 19 such units for this batch, five of them below 3e-5)
  flip/pre             their fp64 pre-activations
  flip/<i>/<name>      the change of every gated gradient that lies below that ReLU when unit i's decision is flipped
                       (gradient with the flipped mask minus the stored gradient; stored like grad/<name>: every `stride`-th
                       entry of the large tensors).  The gradients are LINEAR in the mask, so any set of flipped units is the
                       sum of their rows: the test evaluates the oracle along the HIP path's dense decisions exactly, the way
                       tests/hip_decisions.py does for the tower's ReLU / arg-max decisions.

Needs the same ~35 GB of host memory as make_golden_fullsize.py (one fp64 forward pass with its autograd graph kept); the
stored logits / gradients are checked against the regenerated ones before anything is added.  Run in the build container:
    python tests/golden/make_golden_dense_flips.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from oracle import torch_ref as R             # noqa: E402
import make_golden_fullsize as G              # noqa: E402

BELOW_DENSE = ["InceptionV1/Logits/Conv2d_0c_1x1/weights", "InceptionV1/Logits/Conv2d_0c_1x1/biases", "Text/rnn/basic_lstm_cell/kernel",
               "Text/rnn/basic_lstm_cell/bias", "W_fc", "b_fc", "InceptionV1/Mixed_5c/Branch_0/Conv2d_0a_1x1/BatchNorm/beta"]
THRESH = 1e-4


def main():
    path = os.path.join(HERE, G.CFGS["joint"]["file"])
    old = dict(np.load(path))
    cfg = json.loads(str(old["cfg"]))
    params, emb, batch, mask = G.build(cfg)
    t0 = time.time()
    ref = R.DeepSentimentRef(params, emb, "joint", torch.float64)
    logits = ref.forward(batch, torch.tensor(mask))
    total, _ = ref.loss(logits, batch["labels"])
    assert np.abs(logits.detach().numpy() - old["logits"]).max() <= 1e-12 and abs(float(total) - float(old["loss"])) <= 1e-12
    ps = [ref.p[n] for n in BELOW_DENSE]
    strided = lambda t: (lambda f: (f[::cfg["stride"]] if f.size > cfg["big"] else f))(t.detach().numpy().reshape(-1))
    base = torch.autograd.grad(total, ps + [ref.dense_out], retain_graph=True)
    for n, g in zip(BELOW_DENSE, base):
        assert np.abs(strided(g) - old["grad/" + n]).max() <= 1e-6 * np.abs(old["grad/" + n]).max(), n
    g_out = base[-1]                                   # dL / d relu(pre)
    pre = ref.dense_pre.detach()
    idx = torch.nonzero(pre.abs() < THRESH)
    order = torch.argsort(pre[idx[:, 0], idx[:, 1]].abs())
    idx = idx[order]
    print("%d dense units within %.0e of zero (%.1f s for the forward pass)" % (idx.shape[0], THRESH, time.time() - t0), flush=True)
    new = dict(old)
    new["dense_mask"] = np.packbits((pre > 0).numpy())
    new["flip/units"] = idx.numpy().astype(np.int32)
    new["flip/pre"] = pre[idx[:, 0], idx[:, 1]].numpy()
    for i, (b, j) in enumerate(idx.tolist()):
        cot = torch.zeros_like(pre)
        cot[b, j] = -g_out[b, j] if pre[b, j] > 0 else g_out[b, j]          # the unit's decision flipped
        deltas = torch.autograd.grad(ref.dense_pre, ps, grad_outputs=cot, retain_graph=True, allow_unused=True)
        for n, d in zip(BELOW_DENSE, deltas):
            new["flip/%d/%s" % (i, n)] = strided(d).astype(np.float32)
        rel = [float(np.linalg.norm(strided(d)) / max(np.linalg.norm(old["grad/" + n]), 1e-30)) for n, d in zip(BELOW_DENSE, deltas)]
        print("unit (%3d, %3d) pre %+.2e: relative size of its flip on the gated gradients %s" % (b, j, float(pre[b, j]), " ".join("%.1e" % r for r in rel)), flush=True)
    np.savez_compressed(path, **new)
    print("wrote %s (%.2f MB), %.0f s" % (path, os.path.getsize(path) / 1e6, time.time() - t0))


if __name__ == "__main__":
    main()
