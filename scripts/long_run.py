"""Full-size stability run: joint model, B=256, 400 steps over 8 rotating synthetic batches, LR schedule of the
reference (1e-3 * 0.3^epoch with 'epochs' of 100 steps).  `python scripts/long_run.py [f32|bf16|fp8]`: the same run in
each arithmetic configuration; the bf16 / fp8 runs are the documented "does it train" evidence of those configurations
(they follow the fp32 loss curve on the same data)."""
import sys, time
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
net = SentimentNet(mode="joint", nb_emotions=15, rnn_size=512, vocab_size=10000, embedding_dim=300, post_size=32, dtype=dtype)
net.initialize(seed=1)
batches = [to_device(synthetic_batch_numpy(256, 32, 10000, 15, seed=s)) for s in range(8)]
t0 = time.time()
for step in range(400):
    lr = 1e-3 * 0.3 ** (step // 100)
    net.train_step(batches[step % 8], lr)
    if (step + 1) % 25 == 0:
        loss = net.total_loss_value()
        ok = bool(torch.isfinite(net.store.theta).all())
        print("%s step %4d  lr %.2e  total loss %.4f  finite=%s  (%.1f ms/step incl. logging)" % (dtype, step + 1, lr, loss, ok, (time.time() - t0) / (step + 1) * 1e3), flush=True)
acc = 0
for b in batches:
    acc += int((net.predict(b, is_training=True).argmax(1) == b["labels"]).sum())
print("%s: training-set accuracy on the 8 memorised batches (BN batch statistics): %.3f" % (dtype, acc / (8 * 256)))
