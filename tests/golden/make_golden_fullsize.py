#!/usr/bin/env python
"""Generate the FULL-SIZE oracle vectors (VERDICT r02 "missing" #5, SURVEY 8(c) parity protocol):

    tests/golden/joint_step_b256_oracle.npz   BASELINE configs[2]: joint train_deep_sentiment step, B = 256, T = 32,
                                              V = 10 000, D = 300, H = 512, 224x224 images
    tests/golden/image_step_b128_oracle.npz   BASELINE configs[1]: image-only train_image_model step, B = 128

from the fp64 PyTorch-CPU oracle (oracle/torch_ref.py; TensorFlow 1.x cannot be installed here, so these are oracle
regression vectors, not reference outputs).  Run in the build container (needs ~45 GB of host memory and ~10 minutes):

    python tests/golden/make_golden_fullsize.py joint        # or: image

Inputs and weights are NOT stored: they are regenerated from the seeds in `cfg` by `build()` (shared with
tests/test_golden_gpu.py).  Stored: logits, loss (CE + L2), the gradients of the Logits conv, the dense heads, the LSTM
and of FIVE BatchNorm betas at different depths (tensors above 20k entries as every `stride`-th entry + their L2 norm),
and -- for the betas, which sit below ReLU / arg-max decisions -- `spread/<name>`: the relative L2 distance between this
fp64 result and the SAME oracle run in fp32 (the resolution of any fp32-vs-fp64 comparison, scripts/oracle_fp32_spread.py).
"""
import json
import os
import resource
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import tf_semantics as S          # noqa: E402
from oracle import torch_ref as R             # noqa: E402

CFGS = {
    "joint": dict(mode="joint", B=256, T=32, V=10000, D=300, H=512, param_seed=61, batch_seed=23, lr=1e-3, beta_std=0.1,
                  big=20000, stride=64, file="joint_step_b256_oracle.npz"),
    "image": dict(mode="image", B=128, T=32, V=10000, D=300, H=512, param_seed=62, batch_seed=24, lr=1e-3, beta_std=0.1,
                  big=20000, stride=64, file="image_step_b128_oracle.npz"),
}
BETAS = ["InceptionV1/Conv2d_2c_3x3/BatchNorm/beta", "InceptionV1/Mixed_3c/Branch_2/Conv2d_0b_3x3/BatchNorm/beta",
         "InceptionV1/Mixed_4d/Branch_1/Conv2d_0a_1x1/BatchNorm/beta", "InceptionV1/Mixed_5b/Branch_3/Conv2d_0b_1x1/BatchNorm/beta",
         "InceptionV1/Mixed_5c/Branch_0/Conv2d_0a_1x1/BatchNorm/beta"]


def build(cfg, dtype=np.float64):
    """Seeded problem construction shared with tests/test_golden_gpu.py."""
    mode = cfg["mode"]
    rng = np.random.RandomState(cfg["param_seed"])
    params = R.make_params(mode, rng, num_classes=15, im_features_size=256, embed_dim=cfg["D"], rnn_size=cfg["H"],
                           fc_size=512, dtype=dtype)
    for k in params:
        if k.endswith("beta"):
            params[k] = rng.normal(0, cfg["beta_std"], size=params[k].shape).astype(dtype)
    mask = (rng.uniform(size=(cfg["B"], 1024)) < 0.8).astype(np.float64)
    emb = S.synthetic_embedding(cfg["V"], cfg["D"]).astype(dtype) if mode == "joint" else None
    batch = S.synthetic_batch(cfg["B"], cfg["T"], cfg["V"], seed=cfg["batch_seed"])
    return params, emb, batch, mask


def stored_names(grads):
    keep = [n for n in grads if not n.startswith("InceptionV1/") or "/Logits/" in n]
    return keep + [b for b in BETAS if b in grads]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "joint"
    cfg = dict(CFGS[which])
    if len(sys.argv) > 2:                       # dry run at a smaller batch (timing / memory probe): not written
        cfg["B"] = int(sys.argv[2])
    params, emb, batch, mask = build(cfg)
    t0 = time.time()
    ref = R.DeepSentimentRef(params, emb, cfg["mode"], torch.float64)
    out = ref.train_step(batch, cfg["lr"], torch.tensor(mask))
    t1 = time.time()
    names = stored_names(out["grads"])
    g64 = {n: out["grads"][n].numpy().copy() for n in names}
    logits, loss = out["logits"].numpy().copy(), float(out["loss"])
    del ref, out
    print("fp64 step: %.1f s, peak RSS %.1f GB" % (t1 - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)
    ref32 = R.DeepSentimentRef(params, emb, cfg["mode"], torch.float32)
    out32 = ref32.train_step(batch, cfg["lr"], torch.tensor(mask, dtype=torch.float32))
    arrays = dict(cfg=json.dumps(cfg), logits=logits, loss=np.float64(loss),
                  logits_fp32_spread=np.float64(np.abs(out32["logits"].numpy() - logits).max()))
    for n in names:
        g = g64[n]
        spread = float(np.linalg.norm(out32["grads"][n].numpy().astype(np.float64) - g) / max(np.linalg.norm(g), 1e-30))
        arrays["spread/" + n] = np.float64(spread)
        arrays["gradnorm/" + n] = np.float64(np.linalg.norm(g))
        flat = g.reshape(-1)
        arrays["grad/" + n] = (flat[::cfg["stride"]] if flat.size > cfg["big"] else flat).astype(np.float32)
        print("%-70s |g| %.3e  fp32 spread %.2e" % (n, np.linalg.norm(g), spread))
    print("loss %.6f, oracle fp32-vs-fp64 on the logits: %.2e" % (loss, float(arrays["logits_fp32_spread"])))
    if len(sys.argv) > 2:
        return
    path = os.path.join(HERE, cfg["file"])
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.2f MB), %d gradients, %.0f s" % (path, os.path.getsize(path) / 1e6, len(names), time.time() - t0))


if __name__ == "__main__":
    main()
