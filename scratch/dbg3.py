import numpy as np, torch, sys
sys.path.insert(0, '.')
from oracle import tf_semantics as S
from oracle import torch_ref as R
from tumblr_emotions_amd.net import SentimentNet
rng = np.random.RandomState(23)
B = 3
params = R.make_params("image", rng, num_classes=15, dtype=np.float64)
batch = S.synthetic_batch(B, 8, 10, seed=5)
mask = (rng.uniform(size=(B, 1024)) < 0.8).astype(np.float64)
ref = R.DeepSentimentRef(params, None, "image", torch.float64)
# hook intermediate activations of the oracle
acts = {}
orig_cbr = ref._cbr
def cbr(x, scope, stride=1):
    y = orig_cbr(x, scope, stride); y.retain_grad(); acts[scope] = y; return y
ref._cbr = cbr
logits = ref.forward(batch, torch.tensor(mask)); ref.last_mixed_5c.retain_grad()
total, ce = ref.loss(logits, batch["labels"]); total.backward()
net = SentimentNet(mode="image", nb_emotions=15)
net.load_state_dict(params)
dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
net.train_step(dev, 1e-3, dropout_mask=torch.tensor(mask, dtype=torch.float32).cuda())
torch.cuda.synchronize()
eng = net.image
gref = ref.last_mixed_5c.grad.permute(0,2,3,1).numpy()
g = eng.last.dout.cpu().numpy()
e = np.abs(g-gref); sc = np.abs(gref).max()
print('dout 5c: rel err by branch', [e[...,a:b].max()/sc for a,b in [(0,384),(384,768),(768,896),(896,1024)]])
st = eng.last
pre = 'InceptionV1/Mixed_5c/'
def cmp(name, ours, scope):
    r = acts[scope].grad.permute(0,2,3,1).numpy().reshape(ours.shape[0], -1)
    o = ours.cpu().numpy()
    print(name, 'rel err', np.abs(o-r).max()/np.abs(r).max())
cmp('dr1', st.dr1, pre+'Branch_1/Conv2d_0a_1x1')
cmp('dr2', st.dr2, pre+'Branch_2/Conv2d_0a_1x1')
# forward check of reduce outputs
for nm, ours, scope in [('r1', st.r1, pre+'Branch_1/Conv2d_0a_1x1'), ('r2', st.r2, pre+'Branch_2/Conv2d_0a_1x1')]:
    r = acts[scope].detach().permute(0,2,3,1).numpy().reshape(ours.shape[0], -1)
    print(nm, 'fwd err', np.abs(ours.cpu().numpy()-r).max())
prev = st.prev
# grad wrt 5b output
b5 = None
