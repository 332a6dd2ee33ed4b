#!/usr/bin/env python
"""Generate tests/golden/joint_step_b16_oracle.npz: ONE joint training step at B = 16, 224x224 images,
computed by the fp64 PyTorch-CPU oracle (TensorFlow cannot be installed here, so this is an oracle regression
vector, not a reference output).  Run in the build container:

    python tests/golden/make_golden_step.py

Inputs and weights are NOT stored: they are regenerated from the seeds in `cfg` by the same seeded
constructors the test calls (oracle.torch_ref.make_params, oracle.tf_semantics.synthetic_batch).  Stored:
logits, loss, the gradient of EVERY trainable variable (all 57 BatchNorm betas, the six Mixed_5c weight
tensors, Logits, LSTM, heads; tensors above 100k entries as every 8th entry + their L2 norm), the post-Adam
value of every variable up to 100k entries, every moving mean / variance after the step, and -- per variable --
`spread/<name>`: the relative L2 distance between this fp64 result and the SAME oracle run in fp32.  That
spread (~1e-2 on the tower's gradients at any batch size: ReLU / arg-max decisions that flip within fp32
rounding, scripts/oracle_fp32_spread.py) is the resolution of any fp32-vs-fp64 comparison of this step, so the
test gates each variable by 3x its own spread (floor 1e-3); the tight 1e-3 check is the decision-injected
test in tests/test_model_gpu.py.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import tf_semantics as S          # noqa: E402
from oracle import torch_ref as R             # noqa: E402

CFG = dict(B=16, T=12, V=60, D=20, H=32, param_seed=51, batch_seed=17, lr=1e-3, beta_std=0.1, big=100000, stride=8)


def build(cfg, dtype=np.float64):
    """Seeded problem construction shared with tests/test_golden_gpu.py."""
    rng = np.random.RandomState(cfg["param_seed"])
    params = R.make_params("joint", rng, num_classes=15, im_features_size=256, embed_dim=cfg["D"], rnn_size=cfg["H"],
                           fc_size=512, dtype=dtype)
    for k in params:
        if k.endswith("beta"):
            params[k] = rng.normal(0, cfg["beta_std"], size=params[k].shape).astype(dtype)
    mask = (rng.uniform(size=(cfg["B"], 1024)) < 0.8).astype(np.float64)
    emb = S.synthetic_embedding(cfg["V"], cfg["D"]).astype(dtype)
    batch = S.synthetic_batch(cfg["B"], cfg["T"], cfg["V"], seed=cfg["batch_seed"])
    return params, emb, batch, mask


def main():
    cfg = CFG
    params, emb, batch, mask = build(cfg)
    ref = R.DeepSentimentRef(params, emb, "joint", torch.float64)
    out = ref.train_step(batch, cfg["lr"], torch.tensor(mask))
    ref32 = R.DeepSentimentRef(params, emb, "joint", torch.float32)
    out32 = ref32.train_step(batch, cfg["lr"], torch.tensor(mask, dtype=torch.float32))
    arrays = dict(cfg=json.dumps(cfg), logits=out["logits"].numpy(), loss=np.float64(out["loss"]))
    spreads = []
    for n, g in out["grads"].items():
        g64 = g.numpy()
        spread = float((out32["grads"][n].double() - g).norm() / max(float(g.norm()), 1e-30))
        spreads.append((spread, n))
        arrays["spread/" + n] = np.float64(spread)
        arrays["gradnorm/" + n] = np.float64(np.linalg.norm(g64))
        flat = g64.reshape(-1)
        arrays["grad/" + n] = (flat[::cfg["stride"]] if flat.size > cfg["big"] else flat).astype(np.float32)
    for n in ref.trainable:
        w = ref.p[n].detach().numpy().reshape(-1)
        if w.size <= cfg["big"]:
            arrays["adam/" + n] = w.astype(np.float32)
    for n, v in ref.p.items():
        if n.endswith("moving_mean") or n.endswith("moving_variance"):
            arrays["moving/" + n] = v.numpy().astype(np.float32)
    path = os.path.join(HERE, "joint_step_b16_oracle.npz")
    np.savez_compressed(path, **arrays)
    spreads.sort(reverse=True)
    print("wrote %s (%.2f MB), %d gradients" % (path, os.path.getsize(path) / 1e6, len(out["grads"])))
    print("fp32-vs-fp64 spread of the oracle itself: max %.3e (%s), median %.3e"
          % (spreads[0][0], spreads[0][1], float(np.median([s for s, _ in spreads]))))


if __name__ == "__main__":
    main()
