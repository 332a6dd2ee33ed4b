import numpy as np, torch, sys
sys.path.insert(0, '.')
from oracle import tf_semantics as S
from oracle import torch_ref as R
rng = np.random.RandomState(23)
B = 3
params = R.make_params("image", rng, num_classes=15, dtype=np.float64)
for k in params:
    if k.endswith("beta"):
        params[k] = rng.normal(0, 0.1, size=params[k].shape)
batch = S.synthetic_batch(B, 8, 10, seed=5)
mask = (rng.uniform(size=(B, 1024)) < 0.8).astype(np.float64)
outs = []
for dt in (torch.float64, torch.float32):
    ref = R.DeepSentimentRef(params, None, "image", dt)
    outs.append(ref.train_step(batch, 1e-3, torch.tensor(mask, dtype=dt)))
print('logits diff', (outs[0]['logits'] - outs[1]['logits'].double()).abs().max().item(), 'loss diff', outs[0]['loss']-outs[1]['loss'])
rows = []
for n, g in outs[0]['grads'].items():
    g2 = outs[1]['grads'][n].double()
    rows.append(((g-g2).abs().max().item()/max(g.abs().max().item(),1e-12), n))
rows.sort(reverse=True)
for r in rows[:8]: print("%.3e %s" % r)
