import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
batch = to_device(synthetic_batch_numpy(8, 10, 50, seed=1, with_images=True))
outs = []
for graphed in (False, True):
    net = SentimentNet(mode="joint", nb_emotions=15, rnn_size=32, vocab_size=50, embedding_dim=20, post_size=10)
    net.initialize(seed=3)
    if graphed:
        assert net.capture_step(batch)
    losses = []
    for i in range(3):
        net.train_step(batch, 1e-3 * (0.5 ** i))
        losses.append(net.total_loss_value())
    torch.cuda.synchronize()
    outs.append((losses, net.logits.clone(), net.store.theta.clone(), net.store.frozen.clone()))
(l0, z0, th0, fr0), (l1, z1, th1, fr1) = outs
print(l0, l1)
print("logits", float((z0 - z1).abs().max()), "theta", float((th0 - th1).abs().max()), float(((th0 - th1).abs() <= 1e-5).float().mean()),
      "frozen", float((fr0 - fr1).abs().max()))
