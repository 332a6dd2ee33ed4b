"""Full-size stability run: joint model, B=256, 400 steps over 8 rotating synthetic batches, LR schedule of the
reference (1e-3 * 0.3^epoch with 'epochs' of 100 steps)."""
import sys, time
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
net = SentimentNet(mode="joint", nb_emotions=15, rnn_size=512, vocab_size=10000, embedding_dim=300, post_size=32)
net.initialize(seed=1)
batches = [to_device(synthetic_batch_numpy(256, 32, 10000, 15, seed=s)) for s in range(8)]
t0 = time.time()
for step in range(400):
    lr = 1e-3 * 0.3 ** (step // 100)
    net.train_step(batches[step % 8], lr)
    if (step + 1) % 25 == 0:
        loss = net.total_loss_value()
        ok = bool(torch.isfinite(net.store.theta).all())
        print("step %4d  lr %.2e  total loss %.4f  finite=%s  (%.1f ms/step incl. logging)" % (step + 1, lr, loss, ok, (time.time() - t0) / (step + 1) * 1e3), flush=True)
acc = 0
for b in batches:
    acc += int((net.predict(b, is_training=True).argmax(1) == b["labels"]).sum())
print("training-set accuracy on the 8 memorised batches (BN batch statistics): %.3f" % (acc / (8 * 256)))
