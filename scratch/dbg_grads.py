import numpy as np, torch, sys
sys.path.insert(0, '.')
from oracle import tf_semantics as S
from oracle import torch_ref as R
from tumblr_emotions_amd.net import SentimentNet
rng = np.random.RandomState(23)
B = 3
params = R.make_params("image", rng, num_classes=15, dtype=np.float64)
for k in params:
    if k.endswith("beta"):
        params[k] = rng.normal(0, 0.1, size=params[k].shape)
batch = S.synthetic_batch(B, 8, 10, seed=5)
mask = (rng.uniform(size=(B, 1024)) < 0.8).astype(np.float64)
ref = R.DeepSentimentRef(params, None, "image", torch.float64)
net = SentimentNet(mode="image", nb_emotions=15)
net.load_state_dict(params)
out = ref.train_step(batch, 1e-3, torch.tensor(mask))
dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
net.train_step(dev, 1e-3, dropout_mask=torch.tensor(mask, dtype=torch.float32).cuda())
torch.cuda.synchronize()
grads = net.grads_state_dict()
rows = []
for name, g_ref in out["grads"].items():
    g_ref = g_ref.numpy(); g = grads[name].reshape(g_ref.shape)
    scale = max(np.abs(g_ref).max(), 1e-12)
    rows.append((np.abs(g - g_ref).max() / scale, name, scale))
rows.sort(reverse=True)
for r in rows[:25]: print("%.3e  %-60s scale %.3e" % r)
print('...')
for r in rows[-5:]: print("%.3e  %-60s scale %.3e" % r)
name = "InceptionV1/Mixed_5c/Branch_2/Conv2d_0b_3x3/weights"
g_ref = out["grads"][name].numpy(); g = grads[name]
e = np.abs(g - g_ref)
print('err by tap', e.max(axis=(2,3)))
print('err by ci', e.max(axis=(0,1,3))[:48])
print('err by co first 16', e.max(axis=(0,1,2))[:16])
