#!/usr/bin/env python
"""Upper bounds for restructuring the non-MFMA passes: run bench.py with some op wrappers turned into no-ops (results are
garbage, only the step time matters).
  SKIP=apply,bwd_apply,finalize,bwd_finalize,pool_fwd,pool_bwd,bwd_reduce python scripts/microbench/skip_ops.py [bench args]
finer: bwd_apply_1x1 (BatchNorm backward apply of the 1x1 layers only), b3pool_fwd / b3pool_bwd (Branch_3's 3x3/1 pool)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tumblr_emotions_amd import engine_image, ops
import bench
NAMES = {"apply": ["bn_apply_relu"], "bwd_apply": ["bn_bwd_apply"], "finalize": ["bn_finalize"],
         "bwd_finalize": ["bn_bwd_finalize_segs", "bn_bwd_finalize"], "pool_fwd": ["maxpool_fwd", "maxpool_bn_relu_fwd"],
         "pool_bwd": ["maxpool_bwd"], "bwd_reduce": ["bn_bwd_reduce"]}
skip = [k for k in os.environ.get("SKIP", "").split(",") if k]
for k in skip:
    for n in NAMES.get(k, []):
        setattr(ops, n, lambda *a, **kw: None)
if "bwd_apply_1x1" in skip:
    cur = {"k": 0}
    orig_bwd, orig_apply = engine_image.ConvBN.backward, ops.bn_bwd_apply

    def bwd(self, *a, **kw):
        cur["k"] = self.k if not self.trainable else 0
        return orig_bwd(self, *a, **kw)

    def apply(*a, **kw):
        if cur["k"] != 1:
            return orig_apply(*a, **kw)
    engine_image.ConvBN.backward, ops.bn_bwd_apply = bwd, apply
if "b3pool_fwd" in skip:
    engine_image.MixedStage._pool_fwd = lambda self: None
if "b3pool_bwd" in skip:
    orig_pb = ops.maxpool_bwd

    def pb(dout, argmax, dx, accumulate, B, H, W, C_, k, stride, padding):
        if not (k == 3 and stride == 1):
            return orig_pb(dout, argmax, dx, accumulate, B, H, W, C_, k, stride, padding)
    ops.maxpool_bwd = pb
bench.main()
