#!/usr/bin/env python
"""Soak of the persistent LSTM's hand-off forms: text-only training for N steps; prints the final loss and a digest of
every parameter.  Run once as it is (XCD-local launch: plain stores into the exchange ring where the measured placement
allows) and once with DS_LSTM_XCD=0 (spread row groups: write-through): the digests must be equal -- a lost or stale
hand-off would show as a different digest, a NaN or a time-out from ds_lstm_seq_status.
    python scripts/lstm_soak.py [batch] [steps]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
net = SentimentNet(mode="text", nb_emotions=15, rnn_size=512, vocab_size=10000, embedding_dim=300, post_size=32)
net.initialize(seed=3)
batches = [to_device(synthetic_batch_numpy(B, 32, 10000, 15, seed=s, with_images=False)) for s in range(4)]
for step in range(steps):
    net.train_step(batches[step % 4], 1e-3)
    if (step + 1) % 250 == 0:
        loss = net.total_loss_value()              # synchronises and checks the LSTM status words
        assert bool(torch.isfinite(net.store.theta).all())
torch.cuda.synchronize()
digest = hashlib.sha256(net.store.theta.cpu().numpy().tobytes()).hexdigest()
print("B=%d steps=%d XCD=%s loss=%.6f digest=%s" % (B, steps, os.environ.get("DS_LSTM_XCD", "1"), net.total_loss_value(), digest[:16]))
