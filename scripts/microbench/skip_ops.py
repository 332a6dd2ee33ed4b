#!/usr/bin/env python
"""Upper bounds for restructuring the non-MFMA passes: run bench.py with some op wrappers turned into no-ops (results are
garbage, only the step time matters).   SKIP=apply,bwd_apply,finalize,bwd_finalize,pool_fwd,pool_bwd python scripts/microbench/skip_ops.py [bench args]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tumblr_emotions_amd import ops
import bench
NAMES = {"apply": ["bn_apply_relu"], "bwd_apply": ["bn_bwd_apply"], "finalize": ["bn_finalize"],
         "bwd_finalize": ["bn_bwd_finalize_segs", "bn_bwd_finalize"], "pool_fwd": ["maxpool_fwd"], "pool_bwd": ["maxpool_bwd"],
         "bwd_reduce": ["bn_bwd_reduce"]}
for k in os.environ.get("SKIP", "").split(","):
    for n in NAMES.get(k, []):
        setattr(ops, n, lambda *a, **kw: None)
bench.main()
