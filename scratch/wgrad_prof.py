import sys, collections
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd import ops
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
net = SentimentNet(mode="joint", nb_emotions=15, rnn_size=512, vocab_size=10000, embedding_dim=300, post_size=32, concurrent_towers=False, train_all=True, trainable_embedding=True)
net.initialize(seed=1)
batch = to_device(synthetic_batch_numpy(256, 32, 10000, 15, seed=0))
for _ in range(2): net.train_step(batch, 1e-3)
rec = []
orig = ops.WgradPlan.run
def run(self, *a):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); orig(self, *a); e.record(); rec.append((s, e, self))
ops.WgradPlan.run = run
NS = 3
for _ in range(NS): net.train_step(batch, 1e-3)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for s, e, p in rec:
    d = p.d
    k = (d.N*d.OH*d.OW, d.Cin, d.Cout, d.KH*d.KW, p.ws_bytes)
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += s.elapsed_time(e)
tot = 0
for k, (c, ms) in agg.items():
    fl = 2.0*k[0]*k[1]*k[2]*k[3]
    tiles = k[3]*((k[1]+127)//128)*((k[2]+127)//128)
    splits = max(1, k[4]//(4*k[1]*k[2]*k[3]))
    print("M=%6d Cin=%4d Cout=%4d taps=%d tiles=%3d splits=%3d calls/step=%d  %8.1f us  %6.1f TF" % (k[0],k[1],k[2],k[3],tiles,splits,c//NS,1e3*ms/c,fl/(ms/c*1e-3)/1e12))
    tot += ms/NS
print("total %.3f ms/step" % tot)
