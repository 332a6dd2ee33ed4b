"""Does grid quantisation (tiles per workgroup / workgroups per CU) explain the slow small-M layers?"""
import sys, ctypes as C
sys.path.insert(0, '.')
import torch
from tumblr_emotions_amd import ops, _lib
lib = _lib.load()
def bench(N, H, W, Ci, Co, k, kc):
    if kc: plan = ops.ConvPlan(N,H,W,Ci,Ci,k,k,1,Co,Co,Ci*Co,Ci,1,flip=1)
    else:  plan = ops.ConvPlan(N,H,W,Ci,Ci,k,k,1,Co,Co,Ci*Co,1,Co)
    x = torch.randn(N*H*W, Ci, device='cuda'); w = torch.randn(k*k*Ci*Co, device='cuda')*0.05; z = torch.empty(N*H*W, Co, device='cuda')
    def run(): plan.run(ops._p(x), ops._p(w), ops._p(z))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/20
    return ms*1e3, plan.alg_flops/ms/1e9
for (H, Ci, Co, k, kc) in [(7,384,192,3,1),(7,192,384,3,0),(14,256,128,3,1),(14,128,256,3,0),(14,512,288,1,1),(28,192,128,3,1)]:
    for N in (256, 248, 240, 192, 128):
        M = N*H*H
        for tile in ((0,0),(1,1),(1,2)):
            lib.ds_conv_set_tile(*tile)
            us, tf = bench(N,H,H,Ci,Co,k,kc)
            print("H=%2d Cin=%3d Cout=%3d k=%d kc=%d N=%3d M=%6d rt=%5.1f tile %s: %7.1f us %6.1f TF" % (H,Ci,Co,k,kc,N,M,M/128,tile,us,tf))
lib.ds_conv_set_tile(0,0)
