import ctypes as C, sys, os, torch
sys.path.insert(0, '.')
from tumblr_emotions_amd import _lib, ops
from tumblr_emotions_amd._lib import ConvDesc
LAYERS = {  # name: (N,H,W,Cin,KH,Cout, kcontig)
 'conv2c_fwd': (256,56,56,64,3,192,False), 'conv2c_dgrad': (256,56,56,192,3,64,True),
 '3b_b1_fwd': (256,28,28,96,3,128,False), '4e_fused_fwd': (256,14,14,512,1,288,False),
 '4f_dgrad1x1': (256,14,14,448,1,528,True), '5c_b1_fwd': (256,7,7,192,3,384,False), 'lstm_rec': None}
def bench(libpath, name, cfgs):
    lib = C.CDLL(libpath)
    lib.ds_conv_igemm.restype = C.c_int
    lib.ds_conv_igemm.argtypes = [C.POINTER(ConvDesc)] + [C.c_void_p]*7
    N,H,W,Ci,k,Co,kc = LAYERS[name]
    d = ConvDesc(); d.N,d.H,d.W,d.Cin,d.ldx = N,H,W,Ci,Ci; d.KH=d.KW=k; d.stride=1; d.pad_t=d.pad_l=k//2; d.OH,d.OW=H,W
    d.Cout,d.ldz = Co,Co
    if kc: d.w_tap_stride, d.w_n_stride, d.w_k_stride, d.flip = Ci*Co, Ci, 1, 1   # dgrad view of a [k,k,Co,Ci] tensor
    else:  d.w_tap_stride, d.w_n_stride, d.w_k_stride, d.flip = Ci*Co, 1, Co, 0
    x = torch.randn(N*H*W, Ci, device='cuda'); w = torch.randn(k*k*Ci*Co, device='cuda')*0.05; z = torch.empty(N*H*W, Co, device='cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run(): assert lib.ds_conv_igemm(C.byref(d), x.data_ptr(), w.data_ptr(), z.data_ptr(), None, None, None, st) == 0
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/10
    fl = 2.0*N*H*W*Co*k*k*Ci
    return ms, fl/ms/1e9
if __name__ == '__main__':
    lib, cfg = sys.argv[1], sys.argv[2]
    os.environ['DS_CONV_CFG'] = cfg
    for name in ['conv2c_fwd','conv2c_dgrad','3b_b1_fwd','4e_fused_fwd','4f_dgrad1x1','5c_b1_fwd']:
        ms, tf = bench(lib, name, cfg)
        print("%-14s cfg %-5s %-22s %8.3f ms %7.1f TF" % (name, cfg, os.path.basename(lib), ms, tf))
