import os, sys, time, ctypes as C
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
torch.cuda.init(); torch.zeros(1, device='cuda')
hip = C.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32*i)) & 0xffffffff for i in range(8)])
    s = C.c_void_p()
    r = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert r == 0, r
    return torch.cuda.ExternalStream(s.value)
net = SentimentNet(mode="joint", nb_emotions=15, rnn_size=512, vocab_size=10000, embedding_dim=300, post_size=32)
net.initialize(seed=1)
batch = to_device(synthetic_batch_numpy(256, 32, 10000, 15, seed=0))
def run(tag):
    for _ in range(5): net.train_step(batch, 1e-3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): net.train_step(batch, 1e-3)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s: %.3f ms/step" % (tag, dt/20*1e3), flush=True)
run("default text stream")
for ncu, pattern in [(32, "low"), (64, "low"), (128, "low"), (32, "spread"), (64, "spread"), (16, "spread")]:
    if pattern == "low": bits = (1 << ncu) - 1
    else:
        step = 256 // ncu; bits = 0
        for i in range(ncu): bits |= 1 << (i*step)
    net.text_stream = masked_stream(bits)
    run("text tower on %d CUs (%s)" % (ncu, pattern))
