#!/bin/bash
R=$(cd $(dirname $0)/.. && pwd)
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py -q -s -k "dz_storage or z_storage or bf16_multiply or fp8" 2>&1 | grep -v "^$" | tail -10 | cut -c1-200
python - <<'PY' 2>&1 | grep -v amdgpu
import torch
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
for dt in ("bf16", "fp8"):
    net = SentimentNet(mode="joint", nb_emotions=15, rnn_size=32, vocab_size=50, embedding_dim=20, post_size=10, dtype=dt)
    net.initialize(seed=7)
    net.train_step(to_device(synthetic_batch_numpy(8, 10, 50, seed=5)), 1e-3)
    print(dt, "dz16 layers", sum(1 for l in net.image.layers if getattr(l, "dz16", None) is not None), "z16 layers", sum(1 for l in net.image.layers if l.z16), "of", len(net.image.layers))
PY
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do echo "bf16 $(run --dtype bf16)"; echo "fp8 $(run --dtype fp8)"; echo "bf16_B128 $(run --dtype bf16 --batch 128)"; done
