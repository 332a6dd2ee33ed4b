#!/usr/bin/env python
"""joint_fp8_b128_oracle.npz: BASELINE configs[4]'s per-GPU share (joint model, B = 128 = 1024 / 8, T = 32, V = 10 000,
D = 300, H = 512) through the fp64 oracle with the fp8 configuration's multiplies EMULATED the way the build ships them
(DeepSentimentRef.conv_multiply = "fp8_auto": ds_conv_fp8's e4m3 / e5m2 quantisation with per-tensor power-of-two scales on
the layers the launch rule gives to fp8, bf16 rounding on the others, fp32-exact everything else), and with exact
multiplies for comparison.  Run in the build container (about 20 GB and a few minutes):

    python tests/golden/make_golden_fp8.py

Inputs and weights are regenerated from the seeds by make_golden_fullsize.build (shared with tests/test_golden_gpu.py).
Stored: logits and loss of both runs, and for the emulated run the gradients of the heads / LSTM / Logits (as in the
full-size vectors)."""
import os
import resource
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from oracle import torch_ref as R                      # noqa: E402
from make_golden_fullsize import build, stored_names   # noqa: E402

CFG = dict(mode="joint", B=128, T=32, V=10000, D=300, H=512, param_seed=63, batch_seed=25, lr=1e-3, beta_std=0.1,
           big=20000, stride=64, file="joint_fp8_b128_oracle.npz")


def main():
    cfg = dict(CFG)
    if len(sys.argv) > 1:
        cfg["B"] = int(sys.argv[1])                    # dry run at a smaller batch: not written
    params, emb, batch, mask = build(cfg)
    res = {}
    for kind in ("fp8_auto", "f32"):
        t0 = time.time()
        ref = R.DeepSentimentRef(params, emb, "joint", torch.float64)
        ref.conv_multiply = kind
        out = ref.train_step(batch, cfg["lr"], torch.tensor(mask))
        print("%s step: %.1f s, peak RSS %.1f GB" % (kind, time.time() - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6),
              flush=True)
        res["logits/" + kind] = out["logits"].numpy().copy()
        res["loss/" + kind] = np.float64(out["loss"])
        if kind == "fp8_auto":
            for n in stored_names(out["grads"]):
                g = out["grads"][n].numpy()
                res["norm/" + n] = np.float64(np.linalg.norm(g))
                res["grad/" + n] = g.reshape(-1)[::cfg["stride"]].copy() if g.size > cfg["big"] else g.copy()
        del ref, out
    print("max |logits(fp8_auto) - logits(f32)| = %.3f, loss %.4f vs %.4f" % (
        np.abs(res["logits/fp8_auto"] - res["logits/f32"]).max(), res["loss/fp8_auto"], res["loss/f32"]))
    if cfg["B"] == CFG["B"]:
        np.savez_compressed(os.path.join(HERE, cfg["file"]), cfg=np.array(repr(cfg)), **res)
        print("wrote", cfg["file"])


if __name__ == "__main__":
    main()
