#!/bin/bash
# Branch_3's dgrad output in bf16 storage (DS_DPOOLED16): step check + A/B
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
mkdir -p gpurun_out/r06ze
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -8
import torch, numpy as np
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
batch = to_device(synthetic_batch_numpy(32, 10, 50, seed=5))
for dt in ("bf16", "fp8"):
    res = []
    for on in (True, False):
        net = SentimentNet(mode="joint", nb_emotions=15, rnn_size=32, vocab_size=50, embedding_dim=20, post_size=10, dtype=dt)
        net.image.dpooled16 = on
        net.initialize(seed=7)
        net.train_step(batch, 1e-3)
        torch.cuda.synchronize()
        n16 = sum(1 for st in net.image.stages if getattr(st, "dpooled16", False))
        res.append((net.logits.detach().clone(), net.total_loss_value(), net.grads_state_dict(), n16))
    rels = [np.linalg.norm(res[0][2][k].astype(np.float64) - g) / max(np.linalg.norm(g), 1e-30) for k, g in res[1][2].items()]
    print(dt, "blocks:", res[0][3], res[1][3], "logits equal:", bool(torch.equal(res[0][0], res[1][0])), "gradient rel L2 median %.3e worst %.3e" % (np.median(rels), max(rels)))
PY
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do for e in 1 0; do echo "bf16 dpooled16=$e $(DS_DPOOLED16=$e run --dtype bf16)"; echo "bf16_B128 dpooled16=$e $(DS_DPOOLED16=$e run --dtype bf16 --batch 128)"; done; done > gpurun_out/r06ze/ab.txt 2>&1
sort gpurun_out/r06ze/ab.txt
