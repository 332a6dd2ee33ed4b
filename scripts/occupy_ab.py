#!/usr/bin/env python
"""What does a co-resident persistent launch cost the image tower?  The image-only training step (B = 256 by default) timed with a
diagnostic kernel (scripts/microbench/occupy.hip) sitting on WGS CUs' worth of workgroups for the whole timed region.
usage: occupy_ab.py [batch] [dtype]"""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from tumblr_emotions_amd import streams
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
DT = sys.argv[2] if len(sys.argv) > 2 else "f32"
lib = ctypes.CDLL(os.path.join(R, "scripts", "microbench", "liboccupy.so"))
lib.occupy_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
torch.cuda.set_device(0)
streams.reserve()
net = SentimentNet(mode="image", nb_emotions=15, im_features_size=256, rnn_size=512, fc_size=512, vocab_size=10000,
                   embedding_dim=300, post_size=32, dropout_keep_prob=0.8, dtype=DT)
net.initialize(seed=1)
batch = to_device(synthetic_batch_numpy(B, 32, 10000, 15, seed=0, with_images=True), "cuda", 0, 1)
for _ in range(8):
    net.train_step(batch, 1e-3)
torch.cuda.synchronize()
occ = torch.cuda.Stream()
word = torch.zeros(64, dtype=torch.int32, device="cuda")
sink = torch.zeros(4096, device="cuda")
STEPS = 30


def timed(wgs, lds, mode):
    torch.cuda.synchronize()
    if wgs:
        est = STEPS * 16.0 * (B / 256.0 + 0.3) * 2.0          # long enough to cover the timed steps (ms)
        rc = lib.occupy_launch(wgs, lds, est, mode, word.data_ptr(), sink.data_ptr(), occ.cuda_stream)
        assert rc == 0, rc
        time.sleep(0.02)                                       # the occupant is resident before the first step
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(STEPS):
        net.train_step(batch, 1e-3)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / STEPS
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    left = time.perf_counter() - t0
    return ms, left


CASES = [("none", 0, 0, 0),
         ("32 wg sleep 84 KB", 32, 84 * 1024, 0), ("32 wg sleep 0 KB", 32, 0, 0), ("32 wg sleep 160 KB", 32, 160 * 1024, 0),
         ("32 wg busy 84 KB", 32, 84 * 1024, 1), ("32 wg poll 84 KB", 32, 84 * 1024, 2),
         ("64 wg sleep 84 KB", 64, 84 * 1024, 0), ("64 wg busy 84 KB", 64, 84 * 1024, 1),
         ("16 wg busy 84 KB", 16, 84 * 1024, 1), ("8 wg busy 84 KB", 8, 84 * 1024, 1)]
for rep in range(2):
    for name, wgs, lds, mode in CASES:
        ms, left = timed(wgs, lds, mode)
        print("B=%d %s | %-20s | %.3f ms/step  (occupant outlived the steps by %.0f ms)" % (B, DT, name, ms, left * 1e3), flush=True)
