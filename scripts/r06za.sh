#!/bin/bash
# z of the frozen Mixed-block layers in bf16 storage (DS_Z16=1): does the 16-bit step run, what does it buy
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
mkdir -p gpurun_out/r06za
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -12
import torch, numpy as np
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
batch = to_device(synthetic_batch_numpy(32, 10, 50, seed=5))
res = []
for on in (True, False):
    net = SentimentNet(mode="joint", nb_emotions=15, rnn_size=32, vocab_size=50, embedding_dim=20, post_size=10, dtype="bf16")
    net.image.z16 = on
    net.initialize(seed=7)
    net.train_step(batch, 1e-3)
    torch.cuda.synchronize()
    n16 = sum(1 for l in net.image.layers if l.z16)
    res.append((net.logits.detach().clone(), net.total_loss_value(), net.grads_state_dict(), n16))
print("layers with bf16 z:", res[0][3], res[1][3])
print("max|dlogits| %.3e  |dloss| %.3e" % (float((res[0][0] - res[1][0]).abs().max()), abs(res[0][1] - res[1][1])))
rels = [np.linalg.norm(res[0][2][k].astype(np.float64) - g) / max(np.linalg.norm(g), 1e-30) for k, g in res[1][2].items()]
print("gradient rel L2 median %.3e worst %.3e" % (np.median(rels), max(rels)))
PY
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('final_loss'))"; }
for i in 1 2 3; do for e in 1 0; do echo "bf16 z16=$e $(DS_Z16=$e run --dtype bf16)"; echo "bf16_B128 z16=$e $(DS_Z16=$e run --dtype bf16 --batch 128)"; done; done > gpurun_out/r06za/ab.txt 2>&1
cat gpurun_out/r06za/ab.txt | sort
