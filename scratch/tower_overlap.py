"""When does the text tower run inside an UNPROFILED joint step?  HIP events around the text tower's forward / backward
(on its stream) and the image tower's (on main), relative to the step's first event."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import streams
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
streams.reserve()
net = SentimentNet(mode="joint", nb_emotions=15, im_features_size=256, rnn_size=512, fc_size=512, vocab_size=10000,
                   embedding_dim=300, post_size=32, dropout_keep_prob=0.8)
net.initialize(seed=1)
batch = to_device(synthetic_batch_numpy(256, 32, 10000, 15, seed=0), "cuda", 0, 1)
marks = {}
def ev(name):
    e = torch.cuda.Event(enable_timing=True); e.record(torch.cuda.current_stream()); marks[name] = e
def wrap(obj, meth, tag):
    f = getattr(obj, meth)
    def g(*a, **k):
        ev(tag + "_begin"); r = f(*a, **k); ev(tag + "_end"); return r
    setattr(obj, meth, g)
wrap(net.text, "forward", "text_fwd"); wrap(net.text, "backward", "text_bwd")
wrap(net.image, "forward", "image_fwd"); wrap(net.image, "backward", "image_bwd")
for it in range(12):
    marks.clear()
    ev("step")
    net.train_step(batch, 1e-3)
    ev("step_end")
torch.cuda.synchronize()
t0 = marks["step"]
for k in ("image_fwd_begin", "text_fwd_begin", "text_fwd_end", "image_fwd_end", "text_bwd_begin", "image_bwd_begin", "text_bwd_end", "image_bwd_end", "step_end"):
    print("%-18s %7.3f ms" % (k, t0.elapsed_time(marks[k])))
