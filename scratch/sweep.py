"""Sweep tile configs over every distinct conv/GEMM shape of the joint step at B=256."""
import sys, collections, ctypes as C
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd import ops, _lib
from tumblr_emotions_amd.net import SentimentNet
from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
B = 256
net = SentimentNet(mode="joint", nb_emotions=15, rnn_size=512, vocab_size=10000, embedding_dim=300, post_size=32)
net.initialize(seed=1)
batch = to_device(synthetic_batch_numpy(B, 32, 10000, 15, seed=0))
net.train_step(batch, 1e-3)
plans = collections.OrderedDict()
class T(ops.ConvTimer):
    def begin(self): pass
    def end(self, plan):
        d = plan.d
        key = (d.N*d.OH*d.OW, d.Cout, d.Cin, d.KH*d.KW, d.w_k_stride == 1, d.flags, d.fold_cin)
        if key not in plans: plans[key] = [plan, 0]
        plans[key][1] += 1
ops.CONV_TIMER = T(); net.train_step(batch, 1e-3); ops.CONV_TIMER = None
torch.cuda.synchronize()
lib = _lib.load()
big = torch.randn(900_000_000, device="cuda")   # generic operand storage
w = torch.randn(4_000_000, device='cuda') * 0.05
bias = torch.zeros(4096, device='cuda')
CFGS = [(1,1,1),(1,1,2),(3,1,2),(4,1,1),(4,1,2),(4,1,3)]
rows = []
for key, (plan, count) in plans.items():
    M, N, K, taps, kc, flags, fold = key
    d = plan.d
    saved = d.flags
    d.flags = saved & ~(ops.DS_EPI_MASK)        # mask source not needed for timing
    x_need = d.N*d.H*d.W*d.ldx
    z_off = ((x_need + 1023)//1024)*1024
    stats = big[z_off + M*d.ldz + 4096:]
    res = {}
    for cfg in [(0,0,0)] + CFGS:
        lib.ds_conv_set_path(cfg[0]); lib.ds_conv_set_tile(cfg[1], cfg[2])
        def run(): plan.run(C.c_void_p(big.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(big.data_ptr()+4*z_off), bias=C.c_void_p(bias.data_ptr()), stats=C.c_void_p(stats.data_ptr()))
        run(); torch.cuda.synchronize()
        reps = 5 if plan.alg_flops > 1e10 else 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        res[cfg] = e0.elapsed_time(e1)/reps
    d.flags = saved
    best = min(CFGS, key=lambda c: res[c])
    rows.append((res[(0,0,0)]*count, key, count, res, best))
lib.ds_conv_set_tile(0,0); lib.ds_conv_set_path(0)
rows.sort(key=lambda r: -r[0])
tot_auto = sum(r[0] for r in rows); tot_best = sum(r[3][r[4]]*r[2] for r in rows)
print("%9s %5s %5s %4s %2s %3s | %8s | %s | best" % ("M","N","K","taps","kc","cnt","auto_us", " ".join("%s%d,%d" % ("ALDGW"[c[0]], c[1], c[2]) for c in CFGS)))
for t, key, count, res, best in rows:
    print("%9d %5d %5d %4d %2d %3d | %8.1f | %s | %s%d,%d %.1f" % (key[0], key[1], key[2], key[3], key[4], count, res[(0,0,0)]*1e3,
          " ".join("%6.0f" % (res[c]*1e3) for c in CFGS), "ALDGW"[best[0]], best[1], best[2], res[best]*1e3))
print("total auto %.3f ms/step   total best-per-shape %.3f ms/step" % (tot_auto, tot_best))
for c in CFGS: print("all %s%d,%d: %.3f" % ("ALDGW"[c[0]], c[1], c[2], sum(r[3][c]*r[2] for r in rows)))
