"""Per-phase cycle counts of the register-staged conv kernel's K-step (s_memtime stamps, DS_EXP=8 build)."""
import ctypes as C, sys, os, torch
sys.path.insert(0, '.')
from tumblr_emotions_amd._lib import ConvDesc
lib = C.CDLL(sys.argv[1])
lib.ds_conv_igemm.restype = C.c_int
lib.ds_conv_igemm.argtypes = [C.POINTER(ConvDesc)] + [C.c_void_p]*7
lib.ds_conv_set_path(1)
LAYERS = {'conv2c_dgrad': (256,56,56,192,3,64,True), 'conv2c_fwd': (256,56,56,64,3,192,False), '4e_fused_fwd': (256,14,14,512,1,288,False), '3b_b1_dgrad': (256,28,28,128,3,96,True)}
for name, (N,H,W,Ci,k,Co,kc) in LAYERS.items():
    d = ConvDesc(); d.N,d.H,d.W,d.Cin,d.ldx = N,H,W,Ci,Ci; d.KH=d.KW=k; d.stride=1; d.pad_t=d.pad_l=k//2; d.OH,d.OW=H,W
    d.Cout,d.ldz = Co,Co
    if kc: d.w_tap_stride, d.w_n_stride, d.w_k_stride, d.flip = Ci*Co, Ci, 1, 1
    else:  d.w_tap_stride, d.w_n_stride, d.w_k_stride, d.flip = Ci*Co, 1, Co, 0
    x = torch.randn(N*H*W, Ci, device='cuda'); w = torch.randn(k*k*Ci*Co, device='cuda')*0.05; z = torch.empty(N*H*W, Co, device='cuda')
    tbuf = torch.zeros(40000*4*8, device='cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run(): assert lib.ds_conv_igemm(C.byref(d), x.data_ptr(), w.data_ptr(), z.data_ptr(), None, None, tbuf.data_ptr(), st) == 0
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    t = tbuf.view(-1, 8).cpu()
    t = t[t[:, 4] > 0]
    ks = t[:, 4].sum().item()
    ph = (t[:, :4].sum(0) / ks).tolist()
    tot = sum(ph)
    print("%-14s %.3f ms | per K-step (cycles of the 100 MHz?/shader counter): loads-issue %.0f  frags+MFMA %.0f  vmcnt+ds_write %.0f  barrier %.0f  = %.0f | K-steps/wave %.0f, kernel cycles/wave %.0f, in-loop share %.2f" % (
        name, e0.elapsed_time(e1), ph[0], ph[1], ph[2], ph[3], tot, ks / len(t), t[:, 5].mean().item(), (tot * ks / len(t)) / t[:, 5].mean().item()))
