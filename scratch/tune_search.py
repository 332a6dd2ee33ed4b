"""Brute-force (tile_nt, grid_x) per distinct conv/GEMM shape of the joint step at B=256."""
import sys, collections, ctypes as C, json, math
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
        key = (d.N*d.OH*d.OW, d.Cout, d.Cin, d.KH*d.KW, int(d.w_k_stride == 1), d.flags, d.fold_cin, d.splits)
        if key not in plans: plans[key] = [plan, 0]
        plans[key][1] += 1
ops.CONV_TIMER = T(); net.train_step(batch, 1e-3); ops.CONV_TIMER = None
torch.cuda.synchronize()
lib = _lib.load()
big = torch.randn(900_000_000, device='cuda')
w = torch.randn(4_000_000, device='cuda') * 0.05
bias = torch.zeros(4096, device='cuda')
def timeit(run, flops):
    run(); torch.cuda.synchronize()
    reps = 4 if flops > 1e10 else 12
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
out = {}
tot_auto = tot_best = 0
for key, (plan, count) in plans.items():
    M, N, K, taps, kc, flags, fold, splits = key
    d = plan.d
    saved = d.flags
    d.flags = saved & ~(ops.DS_EPI_MASK)
    x_need = d.N*d.H*d.W*d.ldx
    z_off = ((x_need + 1023)//1024)*1024
    zsz = M*d.ldz*max(1, splits) if splits <= 1 else int(d.z_split_stride)*splits
    stats = big[z_off + zsz + 4096:]
    def run(): plan.run(C.c_void_p(big.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(big.data_ptr()+4*z_off), bias=C.c_void_p(bias.data_ptr()), stats=C.c_void_p(stats.data_ptr()))
    d.tile_nt = 0; d.grid_x = 0
    t_auto = timeit(run, plan.alg_flops)
    best = (t_auto, 0, 0)
    rt = (M + 127)//128
    res = {}
    for nt in (1, 2, 3):
        if N <= 32 and nt > 1: continue
        if N <= 64 and nt > 2: continue
        gy = (N + 32*nt - 1)//(32*nt)
        cap = {1: (4 if kc else 5), 2: 3, 3: 2}[nt]*256
        gmax = max(1, min(rt, cap//(gy*max(1,splits))))
        tmin = math.ceil(rt/gmax)
        cands = set()
        cands.add(0)      # automatic grid for this nt
        for gx in sorted(cands):
            d.tile_nt = nt; d.grid_x = gx
            t = timeit(run, plan.alg_flops)
            res["%d,%d" % (nt, gx)] = round(t, 1)
            if t < best[0]: best = (t, nt, gx)
    d.tile_nt = 0; d.grid_x = 0
    t_auto2 = timeit(run, plan.alg_flops)
    d.flags = saved
    out["%d,%d,%d,%d,%d,%d,%d" % (M, N, K, taps, kc, fold, splits)] = dict(auto=round(min(t_auto, t_auto2), 1), best=round(best[0], 1), nt=best[1], gx=best[2], count=count, all=res)
    tot_auto += min(t_auto, t_auto2)*count; tot_best += best[0]*count
    print("%9d %5d %5d %2d kc%d sp%d x%2d | auto %7.1f best %7.1f (nt %d gx %4d)  rt %d" % (M, N, K, taps, kc, splits, count, min(t_auto, t_auto2), best[0], best[1], best[2], rt), flush=True)
print("total auto %.3f ms  best %.3f ms" % (tot_auto/1e3, tot_best/1e3))
json.dump(out, open("gpurun_out/tune.json", "w"), indent=0)
