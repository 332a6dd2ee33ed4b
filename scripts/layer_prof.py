import sys, collections
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd import ops
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
import os
net = SentimentNet(mode="joint", nb_emotions=15, rnn_size=512, vocab_size=10000, embedding_dim=300, post_size=32, concurrent_towers=os.environ.get("SERIAL") is None)
net.initialize(seed=1)
batch = to_device(synthetic_batch_numpy(B, 32, 10000, 15, seed=0))
for _ in range(2): net.train_step(batch, 1e-3)
class T(ops.ConvTimer):
    def end(self, plan):
        e = torch.cuda.Event(enable_timing=True); e.record(torch.cuda.current_stream())
        self.records.append((self._start, e, plan))
t = T(); ops.CONV_TIMER = t
NS = 3
for _ in range(NS): net.train_step(batch, 1e-3)
torch.cuda.synchronize(); ops.CONV_TIMER = None
agg = collections.OrderedDict()
for s, e, p in t.records:
    d = p.d
    key = (d.N*d.OH*d.OW, d.Cout, d.Cin, d.KH*d.KW, d.w_k_stride == 1, d.flags, d.fold_cin)
    a = agg.setdefault(key, [0, 0.0, p.alg_flops])
    a[0] += 1; a[1] += s.elapsed_time(e)
tot = 0
print("%9s %5s %5s %4s %5s %5s %6s %9s %8s %7s" % ("M","N","K/tap","taps","kcont","flags","calls","us/call","TFLOP/s","ms/step"))
for k, (c, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += ms
    print("%9d %5d %5d %4d %5s %5d %6d %9.1f %8.1f %7.3f" % (k[0], k[1], k[2], k[3], k[4], k[5], c//NS, 1e3*ms/c, fl*c/(ms*1e-3)/1e12, ms/NS))
print("total conv ms/step", tot/NS)
