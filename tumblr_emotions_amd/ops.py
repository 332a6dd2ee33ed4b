"""Thin, typed wrappers over the C ABI (include/ds_kernels.h) taking torch CUDA tensors.

PyTorch is used for device memory and streams only; every FLOP on the path runs in
libds_kernels.so.  All wrappers enqueue on the *current* torch stream (so they are captured by
torch.cuda.graph) and never synchronise.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import (ConvDesc, Segments, SumSegments, DS_EPI_ACCUM, DS_EPI_BIAS, DS_EPI_BNSUMS, DS_EPI_MASK, DS_EPI_RELU,  # noqa: F401
                   DS_EPI_STATS,
                   DS_DTYPE_BF16, DS_DTYPE_F32, DS_FP8_E4M3, DS_FP8_E5M2, DS_CONV_FWD, DS_CONV_DGRAD, DS_ARITH_F32, DS_ARITH_BF16,
                   DS_ARITH_FP8, DS_ARITH_F32X3, DS_FAM_IGEMM, DS_FAM_WINO2, DS_FAM_WINO4, DS_FAM_STEM, DS_FAM_BF16D, DS_FAM_FP8D,
                   DS_FAM_F32X3, DS_FAM_WINO4H, DS_FAM_STEM_POOL, DS_PLAN_STEM_POOL, DS_PLAN_NO_SPLITK, DS_PLAN_NO_WINO4H, DS_PLAN_NO_WINO, DS_PLAN_NO_WINO4, DS_PLAN_NO_STEM_DIRECT, DS_PLAN_NO_BF16_DIRECT, DS_PLAN_ACT16,
                   DS_PLAN_PACKED_RGB, DS_PLAN_FP8_EVERYWHERE, DS_PLAN_FP8_WIDE_RULE)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream.  The raw-stream query is ~10x cheaper than building a
    torch.cuda.Stream object, which matters at ~900 launches per step."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    """Device pointer of a tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("tumblr_emotions_amd kernels need CUDA/HIP tensors; there is no CPU fallback")
    return C.c_void_p(t.data_ptr())


def same_pad(n, k, s):
    """TF SAME geometry: out = ceil(n/s), extra padding goes bottom/right."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2


def make_segments(entries):
    """entries: list of (c_begin, c_end, address, ld[, dtype[, amax address]]) -> Segments struct (dtype: DS_DTYPE_F32
    default, or DS_DTYPE_BF16 for an activation destination kept in 16-bit storage; ld in elements; amax: device word
    that collects max(y) of the segment for an fp8 consumer)."""
    sg = Segments()
    sg.nseg = len(entries)
    for i, e in enumerate(entries):
        c0, c1, ptr, ld = e[:4]
        sg.c_begin[i], sg.c_end[i], sg.ld[i] = c0, c1, ld
        sg.ptr[i] = ptr
        sg.dtype[i] = e[4] if len(e) > 4 else DS_DTYPE_F32
        sg.amax[i] = e[5] if len(e) > 5 else None
    return sg


def act_dtype(t):
    """DS_DTYPE_* of an activation tensor (fp32, or bf16 under 16-bit activation storage)."""
    return DS_DTYPE_BF16 if t.dtype == torch.bfloat16 else DS_DTYPE_F32


class ConvPlan:
    """A ds_conv_desc plus its launch-derived constants, built once per layer."""

    def __init__(self, N, H, W, Cin, ldx, KH, KW, stride, Cout, ldz, w_tap_stride, w_n_stride, w_k_stride,
                 flip=0, fold_cin=0, flags=0, ldmask=0, pad_t=None, pad_l=None, OH=None, OW=None,
                 splits=1, z_split_stride=0, dtype=DS_DTYPE_F32):
        d = ConvDesc()
        d.dtype = dtype
        d.N, d.H, d.W, d.Cin, d.ldx = N, H, W, Cin, ldx
        d.KH, d.KW, d.stride = KH, KW, stride
        if OH is None:
            OH, pt = same_pad(H, KH, stride)
            OW, pl = same_pad(W, KW if not fold_cin else (Cin // fold_cin), stride)
            pad_t = pt if pad_t is None else pad_t
            pad_l = pl if pad_l is None else pad_l
        d.pad_t, d.pad_l, d.OH, d.OW = pad_t, pad_l, OH, OW
        d.Cout, d.ldz = Cout, ldz
        d.w_tap_stride, d.w_n_stride, d.w_k_stride = w_tap_stride, w_n_stride, w_k_stride
        d.flip, d.fold_cin, d.flags, d.ldmask = flip, fold_cin, flags, ldmask
        d.splits, d.z_split_stride = splits, z_split_stride
        self.d = d
        self.M = N * OH * OW
        self.partials = _lib.load().ds_conv_igemm_partials(C.byref(d)) if flags & (DS_EPI_STATS | DS_EPI_BNSUMS) else 0
        d.partials = self.partials      # a launch that would write another count fails instead of corrupting the stats
        # algorithmic FLOPs of one launch (2*M*N*K over the real, unpadded reduction; the folded
        # stem carries a zero 4th input channel that is not counted)
        k_alg = KH * KW * Cin if not fold_cin else KH * (Cin // fold_cin) * 3
        self.alg_flops = 2.0 * self.M * Cout * k_alg

    def enable_bnsums(self, ldy):
        """Conv2DBackpropInput whose result feeds a BatchNorm + ReLU backward: emit that layer's column sums from the
        epilogue (DS_EPI_BNSUMS; y of pixel stride `ldy` goes in as `mask`, the partials come out of `stats`).
        Returns the partial count, or 0 when the launch for this shape cannot carry the flag."""
        if not _lib.load().ds_conv_igemm_bnsums_supported(C.byref(self.d)):
            return 0
        self.d.partials = 0
        self.d.flags |= DS_EPI_BNSUMS
        self.d.ldmask = ldy
        self.partials = _lib.load().ds_conv_igemm_partials(C.byref(self.d))
        self.d.partials = self.partials
        return self.partials

    def run(self, x, w, z, bias=None, mask=None, stats=None, pivot=None):
        t = CONV_TIMER
        if t is not None:
            t.begin()
        _lib.check(_lib.load().ds_conv_igemm(C.byref(self.d), x, w, z, bias, mask, stats, pivot, _stream()),
                   "ds_conv_igemm")
        if t is not None:
            t.end(self)


class ConvTimer:
    """HIP-event timing of every ds_conv_igemm launch on the stream it is launched on (bench.py)."""

    def __init__(self):
        self.records = []      # (start_event, end_event, alg_flops)
        self._start = None

    def begin(self):
        self._start = torch.cuda.Event(enable_timing=True)
        self._start.record(torch.cuda.current_stream())

    def end(self, plan):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        # (launches that also carry a max pool -- the stem with MaxPool_2a inside, the Branch_3 convs with the pooling loader --
        # are marked: their time is conv + pool, their FLOPs the conv's)
        d = getattr(plan, "d", None)
        pooled = getattr(plan, "family", None) == DS_FAM_STEM_POOL or bool(getattr(d, "pool_argmax", None))
        self.records.append((self._start, e, plan.alg_flops, pooled))

    def summary(self, pooled=None):
        """(launches, total_ms, total_flops) -- call after a device synchronise.  pooled: None = every launch, True / False =
        only the launches with / without a max pool inside."""
        recs = [r for r in self.records if pooled is None or r[3] == pooled]
        ms = sum(s.elapsed_time(e) for (s, e, _, _) in recs)
        return len(recs), ms, sum(f for (_, _, f, _) in recs)


CONV_TIMER = None      # set to a ConvTimer to time the dominant kernel


def head_gemm_plan(M, K, N, lda, ldc, w_ld, transposed_w=False, flags=0, ldmask=0, *, device):
    """gemm_plan for the batch x features GEMMs of the heads (Logits, W_fc, W_softmax and their dgrads): with M <= 512 rows
    and K >= 256 a SplitGemm (split-K + fixed-order combine with the epilogue), else the plain plan.  DS_SPLIT_GEMM=0: always
    the plain plan (A/B aid).  `device`: the engine's device (the slabs live beside its tensors, not on the current device)."""
    if M <= 512 and K >= 256 and _lib.tuning_env("DS_SPLIT_GEMM", "1") != "0":
        return SplitGemm(M, K, N, lda, ldc, w_ld, transposed_w, flags, ldmask, 8 if K >= 512 else 4, device)
    return gemm_plan(M, K, N, lda, ldc, w_ld, transposed_w=transposed_w, flags=flags, ldmask=ldmask)


def gemm_plan(M, K, N, lda, ldc, w_ld, transposed_w=False, flags=0, ldmask=0, splits=1, z_split_stride=0,
              dtype=DS_DTYPE_F32):
    """C[M,N] = A[M,K] * W (row-major W[K,N] with row stride w_ld), or * W^T when transposed_w
    (then W is [N,K] row-major): both are read in place.  splits>1: split-K, slab s of partial
    sums at C + s*z_split_stride (the consumer adds the slabs)."""
    if transposed_w:
        return ConvPlan(M, 1, 1, K, lda, 1, 1, 1, N, ldc, 0, w_ld, 1, flags=flags, ldmask=ldmask,
                        pad_t=0, pad_l=0, OH=1, OW=1, splits=splits, z_split_stride=z_split_stride, dtype=dtype)
    return ConvPlan(M, 1, 1, K, lda, 1, 1, 1, N, ldc, 0, 1, w_ld, flags=flags, ldmask=ldmask,
                    pad_t=0, pad_l=0, OH=1, OW=1, splits=splits, z_split_stride=z_split_stride, dtype=dtype)


class SplitGemm:
    """A batch x features GEMM (M <= 512 rows: one or two row tiles) as split-K ds_conv_igemm + ds_slab_epilogue: the
    single launch walks K serially in a handful of workgroups; `splits` slabs of partial sums and a fixed-order combine
    with the epilogue (bias / accumulate / mask / relu) take a third of the time.  Same call as ConvPlan.run."""

    def __init__(self, M, K, N, lda, ldc, w_ld, transposed_w, flags, ldmask, splits, device):
        self.M, self.N, self.ldc, self.flags, self.ldmask, self.splits = M, N, ldc, flags, ldmask, splits
        self.slabs = torch.empty(splits, M, N, device=device)
        self.plan = gemm_plan(M, K, N, lda, N, w_ld, transposed_w=transposed_w, splits=splits, z_split_stride=M * N)
        self.d, self.alg_flops, self.partials = self.plan.d, self.plan.alg_flops, 0

    def run(self, x, w, z, bias=None, mask=None, stats=None, pivot=None):
        if stats is not None or pivot is not None:
            raise ValueError("SplitGemm has no BatchNorm-statistics epilogue (the heads' GEMMs are not followed by BatchNorm)")
        t = CONV_TIMER
        if t is not None:
            t.begin()
        lib = _lib.load()
        _lib.check(lib.ds_conv_igemm(C.byref(self.plan.d), x, w, _p(self.slabs), None, None, None, None, _stream()),
                   "ds_conv_igemm")
        _lib.check(lib.ds_slab_epilogue(_p(self.slabs), self.splits, self.M * self.N, self.N, self.M, self.N, z, self.ldc,
                                        bias, mask, self.ldmask, self.flags, _stream()), "ds_slab_epilogue")
        if t is not None:
            t.end(self)


class WinoPlan:
    """3x3 stride-1 SAME conv through ds_conv_wino (fused Winograd F(2x2,3x3)) or, with f4, ds_conv_wino4
    (F(4x4,3x3): H, W multiples of four); `u` is the transformed filter (`u_elems` floats)."""

    def __init__(self, N, H, W, Cin, ldx, Cout, ldz, flags=0, f4=False):
        self.args = (N, H, W, Cin, ldx, Cout, ldz)
        self.flags = flags
        self.f4 = bool(f4)
        lib = _lib.load()
        self._run, self._name = (lib.ds_conv_wino4, "ds_conv_wino4") if f4 else (lib.ds_conv_wino, "ds_conv_wino")
        self._partials = lib.ds_conv_wino4_partials if f4 else lib.ds_conv_wino_partials
        self.u_elems = (36 if f4 else 16) * Cin * Cout
        self.M = N * H * W
        self.partials = self._partials(N, H, W) if flags & (DS_EPI_STATS | DS_EPI_BNSUMS) else 0
        self.alg_flops = 2.0 * self.M * Cout * 9 * Cin          # the convolution's FLOPs, not Winograd's

    def enable_bnsums(self):
        """As ConvPlan.enable_bnsums (y has the output's pixel stride)."""
        N, H, W = self.args[:3]
        self.flags = (self.flags & ~DS_EPI_STATS) | DS_EPI_BNSUMS      # (the two sum epilogues exclude each other)
        self.partials = self._partials(N, H, W)
        return self.partials

    def set_ldx(self, ldx):
        self.args = self.args[:4] + (ldx,) + self.args[5:]

    def set_ldz(self, ldz):
        self.args = self.args[:6] + (ldz,)

    def run(self, x, u, z, stats=None, pivot=None, ymask=None):
        t = CONV_TIMER
        if t is not None:
            t.begin()
        N, H, W, Cin, ldx, Cout, ldz = self.args
        _lib.check(self._run(x, u, z, stats, pivot, ymask, N, H, W, Cin, ldx, Cout, ldz, self.flags, _stream()), self._name)
        if t is not None:
            t.end(self)


class LayerPlan:
    """One conv layer launch planned BY THE LIBRARY (ds_conv_plan): the engine says what the layer is -- role (forward /
    Conv2DBackpropInput), arithmetic, filter [k][k][w_cin][w_cout], map, strides of x and z, epilogue flags -- and the
    library picks the kernel family (implicit GEMM / wide 1x1, Winograd F(2x2) / F(4x4), packed-RGB stem, register-direct
    bf16 / fp8 / f32x3), sizes the BatchNorm partials and names the prepared filter form.  `d` is the live descriptor
    (ldx, ldz, flags, norm_* / mask_* may change between runs)."""

    def __init__(self, role, arith, options, N, H, W, w_cin, w_cout, k, stride, ldx, ldz, flags=0):
        self.p = _lib.LayerPlanStruct()
        _lib.check(_lib.load().ds_conv_plan(C.byref(self.p), role, arith, options, N, H, W, w_cin, w_cout, k, stride, ldx,
                                            ldz, flags), "ds_conv_plan")
        self.d = self.p.d                    # a view into self.p
        self.family = self.p.family
        self.M = N * self.d.OH * self.d.OW
        self.alg_flops = self.p.alg_flops
        self.u = None                        # the prepared filter (alloc_weights), None: the family reads HWIO in place
        self.wscale = None
        self.io = _lib.ConvIO()
        self.ws_bytes = int(self.p.ws_bytes)     # > 0: run() needs a scratch tensor of that size (set_workspace): split-K Winograd
        self.splitk = int(self.p.splitk)
        self._ws = None
        self._run = _lib.load().ds_conv_run
        self._ref = C.byref(self.p)
        self._io_ref = C.byref(self.io)

    @property
    def partials(self):
        return self.p.partials

    @property
    def x16_ok(self):
        return bool(self.p.x16_ok)

    def alloc_weights(self, device):
        if self.p.w_bytes:
            self.u = torch.empty(self.p.w_bytes, dtype=torch.uint8, device=device)
        if self.p.wscale_floats:
            self.wscale = torch.zeros(self.p.wscale_floats, device=device)
            self.io.wscale = self.wscale.data_ptr()

    def set_workspace(self, t):
        """Scratch for plans with ws_bytes > 0 (private to the stream the plan runs on; at least ws_bytes long)."""
        assert t.numel() * t.element_size() >= self.ws_bytes
        self._ws = t
        self.io.ws, self.io.ws_bytes = t.data_ptr(), t.numel() * t.element_size()

    def prepare(self, w_hwio):
        """Filter -> the form the chosen family reads (no-op for the families that read HWIO in place)."""
        if self.u is not None:
            _lib.check(_lib.load().ds_conv_prepare_weights(self._ref, w_hwio, _p(self.u), _p(self.wscale), _stream()),
                       "ds_conv_prepare_weights")

    def enable_bnsums(self, ldy):
        """Conv2DBackpropInput whose result feeds a BatchNorm + ReLU backward: that layer's column sums from the epilogue
        (y with pixel stride ldy goes in as `mask`).  Returns the partial count, 0 when the chosen kernel cannot."""
        return int(_lib.load().ds_conv_plan_enable_bnsums(self._ref, ldy))

    def norm_supported(self):
        return bool(_lib.load().ds_conv_plan_norm_supported(self._ref))

    def enable_pool3(self, argmax):
        """Forward 1x1 conv behind a 3x3 / 1 SAME max pool (an Inception block's Branch_3): the pool is formed ON LOAD
        (ds_conv_desc.pool_argmax) -- `argmax` [N*H*W, Cin] uint8 receives the winners ds_maxpool_fwd would record and x is
        the pool's INPUT.  Returns False when the chosen kernel cannot (the caller keeps the separate pool pass)."""
        if not _lib.load().ds_conv_plan_enable_pool3(self._ref, _p(argmax)):
            return False
        self.pool_argmax = argmax            # (kept alive: the descriptor holds its address)
        return True

    def finalize_tickets(self):
        """> 0: ds_conv_run can run ds_bn_finalize INSIDE this launch (run(fin=...)); the number of ticket words it needs."""
        return int(_lib.load().ds_conv_plan_finalize_tickets(self._ref))

    def enable_bn_backward_on_load(self, mean, rstd, shift, coef, parts):
        """Conv2DBackpropInput of a 1x1 conv + BatchNorm + ReLU layer straight from z and the activation gradient: the
        layer's ds_bn_bwd_apply pass is formed on load (ds_conv_desc.bnb).  parts: [(c0, c1, address, ld)] of dy.
        Returns False when the chosen kernel cannot (the caller keeps the separate pass)."""
        if not _lib.load().ds_conv_plan_bnb_supported(self._ref) or len(parts) > 3:
            return False
        if any(c0 % 16 for (c0, _, _, _) in parts):
            return False
        b = _lib.BnBwdOnLoad()
        b.mean, b.rstd, b.shift, b.coef = mean.data_ptr(), rstd.data_ptr(), shift.data_ptr(), coef.data_ptr()
        b.nseg = len(parts)
        for i, (c0, c1, ptr, ld) in enumerate(parts):
            assert c0 == (parts[i - 1][1] if i else 0)
            b.c_end[i], b.ld[i], b.dy[i] = c1, ld, ptr
        self.bnb = b                         # (kept alive: the descriptor holds its address)
        self.d.bnb = C.addressof(b)
        return True

    def run(self, x, w_hwio, z, stats=None, pivot=None, mask=None, bias=None, x_amax=None, fin=None):
        """fin: address of a _lib.BnFinalizeInLaunch (ds_bn_finalize inside the launch, finalize_tickets() > 0) or None."""
        t = CONV_TIMER
        if t is not None:
            t.begin()
        io = self.io
        io.stats, io.pivot, io.mask, io.bias, io.x_amax, io.fin = stats, pivot, mask, bias, x_amax, fin
        _lib.check(self._run(self._ref, x, w_hwio if self.u is None else self.u.data_ptr(), z, self._io_ref, _stream()),
                   "ds_conv_run")
        if t is not None:
            t.end(self)


def conv_norm_supported(plan):
    """Would ds_conv_igemm apply ds_conv_desc.norm_rstd / norm_shift (BatchNorm + ReLU on load) for this plan?"""
    return bool(_lib.load().ds_conv_igemm_norm_supported(C.byref(plan.d)))


def wino4_supported(H, W, Cin, Cout):
    return bool(_lib.load().ds_conv_wino4_supported(H, W, Cin, Cout))


def wino4_prefer(N, H, W, Cin, Cout):
    """The library's launch-time model: is ds_conv_wino4 expected to beat ds_conv_wino on this shape?"""
    return bool(_lib.load().ds_conv_wino4_prefer(N, H, W, Cin, Cout))


def wino_transform_weights(w_ptr, u, Cin, Cout, dgrad, f4=False):
    lib = _lib.load()
    if f4:
        _lib.check(lib.ds_wino4_transform_weights(w_ptr, _p(u), Cin, Cout, int(dgrad), _stream()), "ds_wino4_transform_weights")
    else:
        _lib.check(lib.ds_wino_transform_weights(w_ptr, _p(u), Cin, Cout, int(dgrad), _stream()), "ds_wino_transform_weights")


class StemPlan:
    """Conv2d_1a_7x7 through ds_conv_stem: packed RGB input [N, H, W, 3], HWIO weights with `cin_store` input rows."""

    def __init__(self, N, H, W, cin_store, Cout, ldz, bf16=False):
        self.args = (N, H, W, cin_store, Cout, ldz)
        self.bf16 = bf16               # ds_conv_stem_bf16: operands rounded to bf16, bf16 MFMA (the 16-bit configurations)
        self.flags = DS_EPI_STATS      # kept for the common plan interface: statistics are on iff `stats` is passed
        OH, OW = (H + 1) // 2, (W + 1) // 2
        self.M = N * OH * OW
        self.partials = (_lib.load().ds_conv_stem_bf16_partials if bf16 else _lib.load().ds_conv_stem_partials)(N, OH, OW)
        self.alg_flops = 2.0 * self.M * Cout * 147

    def run(self, x, w, z, stats=None, pivot=None):
        t = CONV_TIMER
        if t is not None:
            t.begin()
        N, H, W, cs, Cout, ldz = self.args
        f = _lib.load().ds_conv_stem_bf16 if self.bf16 else _lib.load().ds_conv_stem
        _lib.check(f(x, w, z, stats, pivot, N, H, W, cs, Cout, ldz, _stream()), "ds_conv_stem")
        if t is not None:
            t.end(self)


class Bf16Plan:
    """1x1 / 3x3 conv (or its dgrad) through ds_conv_bf16: register-direct bf16 MFMA, weights pre-converted by
    `weights_to_bf16` into the kernel's K-loop order.  Geometry arguments as ConvPlan's."""

    def __init__(self, N, H, W, Cin, ldx, k, stride, Cout, ldz, flags=0, pad_t=None, pad_l=None, OH=None, OW=None):
        d = ConvDesc()
        d.dtype = DS_DTYPE_BF16
        d.N, d.H, d.W, d.Cin, d.ldx = N, H, W, Cin, ldx
        d.KH, d.KW, d.stride = k, k, stride
        if OH is None:
            OH, pt = same_pad(H, k, stride)
            OW, pl = same_pad(W, k, stride)
            pad_t = pt if pad_t is None else pad_t
            pad_l = pl if pad_l is None else pad_l
        d.pad_t, d.pad_l, d.OH, d.OW = pad_t, pad_l, OH, OW
        d.Cout, d.ldz, d.flags, d.splits = Cout, ldz, flags, 1
        self.d = d
        lib = _lib.load()
        if not lib.ds_conv_bf16_supported(C.byref(d)):
            raise ValueError("ds_conv_bf16 does not take this geometry")
        self.M = N * OH * OW
        self.partials = lib.ds_conv_bf16_partials(C.byref(d)) if flags & DS_EPI_STATS else 0
        self.alg_flops = 2.0 * self.M * Cout * k * k * Cin

    def set_ldx(self, ldx):
        self.d.ldx = ldx

    @property
    def flags(self):
        return self.d.flags

    @flags.setter
    def flags(self, v):
        self.d.flags = v

    def run(self, x, wb, z, stats=None, pivot=None, mask=None):
        t = CONV_TIMER
        if t is not None:
            t.begin()
        _lib.check(_lib.load().ds_conv_bf16(C.byref(self.d), x, wb, z, mask, stats, pivot, _stream()), "ds_conv_bf16")
        if t is not None:
            t.end(self)


class Fp8Plan(Bf16Plan):
    """1x1 / 3x3 conv (or its dgrad) through ds_conv_fp8: register-direct fp8 MFMA with per-tensor power-of-two scales.
    `wq`, `wscale`: the filter converted by `weights_to_fp8`; `x_amax`: device word holding max|x| (`absmax`);
    a_format: DS_FP8_E4M3 for forward activations, DS_FP8_E5M2 for gradients.  Geometry arguments as ConvPlan's."""

    def __init__(self, N, H, W, Cin, ldx, k, stride, Cout, ldz, flags=0, a_format=DS_FP8_E4M3, **kw):
        super().__init__(N, H, W, Cin, ldx, k, stride, Cout, ldz, flags=flags, **kw)
        lib = _lib.load()
        if not lib.ds_conv_fp8_supported(C.byref(self.d)):
            raise ValueError("ds_conv_fp8 does not take this geometry")
        self.a_format = a_format
        self.partials = lib.ds_conv_fp8_partials(C.byref(self.d)) if flags & DS_EPI_STATS else 0

    def run(self, x, wq, z, stats=None, pivot=None, x_amax=None, wscale=None, mask=None):
        t = CONV_TIMER
        if t is not None:
            t.begin()
        _lib.check(_lib.load().ds_conv_fp8(C.byref(self.d), x, x_amax, self.a_format, wq, wscale, z, mask, stats, pivot,
                                           _stream()), "ds_conv_fp8")
        if t is not None:
            t.end(self)


AMAX_FLOATS = 512       # DS_AMAX_FLOATS: a max|.| record = 16 slots, one per 128-byte line; its value = max of the slots
WSCALE_FLOATS = 4 + AMAX_FLOATS


def amax_value(record):
    """Host value of a max|.| record (tests)."""
    return float(record.view(-1)[:AMAX_FLOATS:32].max().item())


def absmax(x, n, out, x_dtype=DS_DTYPE_F32):
    """out (a record of AMAX_FLOATS floats) <- max |x[:n]| on the device (no host sync).  x: tensor (its dtype counts) or
    raw address + x_dtype."""
    if not isinstance(x, C.c_void_p):
        x_dtype = act_dtype(x)
        x = _p(x)
    _lib.check(_lib.load().ds_absmax(x, n, x_dtype, _p(out), _stream()), "ds_absmax")


def weights_fp8_bytes(Cin, Cout, taps, dgrad):
    return int(_lib.load().ds_weights_fp8_bytes(Cin, Cout, taps, int(dgrad)))


def weights_to_fp8(w_ptr, wq, wscale, Cin, Cout, taps, dgrad):
    """HWIO fp32 filter -> ds_conv_fp8's e4m3 weight tensor `wq` + its scale record `wscale` (4 device floats)."""
    _lib.check(_lib.load().ds_weights_to_fp8(w_ptr, _p(wq), _p(wscale), Cin, Cout, taps, int(dgrad), _stream()),
               "ds_weights_to_fp8")


def weights_bf16_bytes(Cin, Cout, taps, dgrad):
    return int(_lib.load().ds_weights_bf16_bytes(Cin, Cout, taps, int(dgrad)))


def weights_to_bf16(w_ptr, wb, Cin, Cout, taps, dgrad):
    """HWIO fp32 filter -> ds_conv_bf16's weight tensor (`wb`: a uint8/bf16 device tensor of weights_bf16_bytes)."""
    _lib.check(_lib.load().ds_weights_to_bf16(w_ptr, _p(wb), Cin, Cout, taps, int(dgrad), _stream()),
               "ds_weights_to_bf16")


class F32x3Plan:
    """1x1 / 3x3 conv (or its dgrad) through ds_conv_f32x3: fp32 products on the bf16 matrix cores (every operand split into
    three bf16 pieces, six MFMAs per eight of the fp32 path; fp32-MFMA accuracy, not bit-identical).  Weights pre-split by
    `weights_to_f32x3`.  Geometry arguments as Bf16Plan's; x is fp32; d.norm_rstd / norm_shift as for ConvPlan (1x1)."""
    f4 = False

    def __init__(self, N, H, W, Cin, ldx, k, stride, Cout, ldz, flags=0, pad_t=None, pad_l=None, OH=None, OW=None):
        d = ConvDesc()
        d.dtype = DS_DTYPE_F32
        d.N, d.H, d.W, d.Cin, d.ldx = N, H, W, Cin, ldx
        d.KH, d.KW, d.stride = k, k, stride
        if OH is None:
            OH, pt = same_pad(H, k, stride)
            OW, pl = same_pad(W, k, stride)
            pad_t = pt if pad_t is None else pad_t
            pad_l = pl if pad_l is None else pad_l
        d.pad_t, d.pad_l, d.OH, d.OW = pad_t, pad_l, OH, OW
        d.Cout, d.ldz, d.flags, d.splits = Cout, ldz, flags, 1
        self.d = d
        lib = _lib.load()
        if not lib.ds_conv_f32x3_supported(C.byref(d)):
            raise ValueError("ds_conv_f32x3 does not take this geometry")
        self.M = N * OH * OW
        self.partials = lib.ds_conv_f32x3_partials(C.byref(d)) if flags & DS_EPI_STATS else 0
        self.partials_for_sums = lib.ds_conv_f32x3_partials(C.byref(d))      # what a DS_EPI_BNSUMS launch would write
        self.alg_flops = 2.0 * self.M * Cout * k * k * Cin

    def set_ldx(self, ldx):
        self.d.ldx = ldx

    def set_ldz(self, ldz):
        self.d.ldz = ldz

    @property
    def flags(self):
        return self.d.flags

    @flags.setter
    def flags(self, v):
        self.d.flags = v

    def run(self, x, wb, z, stats=None, pivot=None, mask=None):
        t = CONV_TIMER
        if t is not None:
            t.begin()
        _lib.check(_lib.load().ds_conv_f32x3(C.byref(self.d), x, wb, z, mask, stats, pivot, _stream()), "ds_conv_f32x3")
        if t is not None:
            t.end(self)


def weights_f32x3_bytes(Cin, Cout, taps, dgrad):
    return int(_lib.load().ds_weights_f32x3_bytes(Cin, Cout, taps, int(dgrad)))


def weights_to_f32x3(w_ptr, wb, Cin, Cout, taps, dgrad):
    """HWIO fp32 filter -> ds_conv_f32x3's weight tensor (three bf16 pieces per weight, K-loop order)."""
    _lib.check(_lib.load().ds_weights_to_f32x3(w_ptr, _p(wb), Cin, Cout, taps, int(dgrad), _stream()), "ds_weights_to_f32x3")


class WgradPlan:
    """dw[tap, ci, co] = sum_m x[pixel(m)+tap, ci] * dz[m, co]; geometry given by a forward ConvPlan-like desc."""

    def __init__(self, N, H, W, Cin, ldx, KH, KW, stride, Cout, lddz, pad_t=None, pad_l=None, OH=None, OW=None,
                 fold_cin=0):
        d = ConvDesc()
        d.fold_cin = fold_cin
        d.N, d.H, d.W, d.Cin, d.ldx = N, H, W, Cin, ldx
        d.KH, d.KW, d.stride = KH, KW, stride
        if OH is None:
            OH, pad_t = same_pad(H, KH, stride)
            OW, pad_l = same_pad(W, KW if not fold_cin else Cin // fold_cin, stride)
        d.pad_t, d.pad_l, d.OH, d.OW = pad_t, pad_l, OH, OW
        d.Cout, d.ldz = Cout, Cout
        self.d = d
        self.lddz = lddz
        self.ws_bytes = int(_lib.load().ds_conv_wgrad_workspace(C.byref(d)))

    def run(self, x, dz, dw, ws, ws_bytes):
        _lib.check(_lib.load().ds_conv_wgrad(C.byref(self.d), x, dz, self.lddz, dw, ws, ws_bytes, _stream()),
                   "ds_conv_wgrad")


def bn_finalize(stats, P, count, C_, beta, eps, decay, mean, rstd, shift, mm, mv, pivot=None):
    _lib.check(_lib.load().ds_bn_finalize(_p(stats), P, count, C_, _p(beta), _p(pivot), eps, decay, _p(mean),
                                          _p(rstd), _p(shift), _p(mm), _p(mv), _stream()), "ds_bn_finalize")


def bn_finalize_centered(stats, P, count, C_, beta, eps, decay, mean, rstd, shift, mm, mv, pivot, mean_c, shift_c):
    """ds_bn_finalize for a layer whose z is stored centred about the pivot: also mean - pivot and the matching shift."""
    _lib.check(_lib.load().ds_bn_finalize_centered(_p(stats), P, count, C_, _p(beta), _p(pivot), eps, decay, _p(mean), _p(rstd),
                                                   _p(shift), _p(mm), _p(mv), _p(mean_c), _p(shift_c), _stream()),
               "ds_bn_finalize_centered")


def bn_apply_relu(z, M, C_, rstd, shift, segs):
    if z.dtype == torch.bfloat16:          # z in bf16 storage (ds_conv_desc.z_dtype)
        _lib.check(_lib.load().ds_bn_apply_relu_z16(_p(z), M, C_, _p(rstd), _p(shift), C.byref(segs), _stream()),
                   "ds_bn_apply_relu_z16")
        return
    _lib.check(_lib.load().ds_bn_apply_relu(_p(z), M, C_, _p(rstd), _p(shift), C.byref(segs), _stream()),
               "ds_bn_apply_relu")


def bn_infer_prepare(beta, mm, mv, eps, C_, rstd, shift):
    _lib.check(_lib.load().ds_bn_infer_prepare(_p(beta), _p(mm), _p(mv), eps, C_, _p(rstd), _p(shift), _stream()),
               "ds_bn_infer_prepare")


def bn_bwd_partials(M, C_):
    return _lib.load().ds_bn_bwd_partials(M, C_)


def bn_bwd_reduce(z, segs, M, C_, mean, rstd, shift, partials, ldz=None, z_dtype=DS_DTYPE_F32):
    """z .. partials: tensors, or raw device addresses (c_void_p) when a column sub-range of a layer is reduced
    (then z_dtype names z's storage: bf16 for a pooled activation in 16-bit storage)."""
    ptr = lambda t: t if isinstance(t, C.c_void_p) else _p(t)
    if not isinstance(z, C.c_void_p):
        z_dtype = act_dtype(z)
    _lib.check(_lib.load().ds_bn_bwd_reduce(ptr(z), C_ if ldz is None else ldz, z_dtype, C.byref(segs), M, C_, ptr(mean),
                                            ptr(rstd), ptr(shift), ptr(partials), _stream()), "ds_bn_bwd_reduce")


def bn_bwd_finalize_segs(sum_segs, M, C_, beta, dbeta, coef):
    _lib.check(_lib.load().ds_bn_bwd_finalize_segs(C.byref(sum_segs), M, C_, _p(beta), _p(dbeta), _p(coef), _stream()),
               "ds_bn_bwd_finalize_segs")


class BnFinalizeJobs:
    """ds_bn_finalize_multi: the finalizes of up to four layers as one launch.  jobs: (stats, P, count, C, beta, pivot, mean,
    rstd, shift, moving_mean, moving_var) tensors / ints; the moving statistics may be None."""

    def __init__(self, jobs):
        self.n = len(jobs)
        self.arr = (_lib.BnFinalizeJob * self.n)()
        self._keep = jobs
        for a, (stats, P, count, C_, beta, pivot, mean, rstd, shift, mm, mv) in zip(self.arr, jobs):
            a.stats, a.P, a.C, a.count = stats.data_ptr(), P, C_, count
            a.beta, a.pivot = beta.data_ptr(), (pivot.data_ptr() if pivot is not None else None)
            a.mean, a.rstd, a.shift = mean.data_ptr(), rstd.data_ptr(), shift.data_ptr()
            a.moving_mean = mm.data_ptr() if mm is not None else None
            a.moving_var = mv.data_ptr() if mv is not None else None

    def run(self, eps, decay):
        _lib.check(_lib.load().ds_bn_finalize_multi(C.cast(self.arr, C.c_void_p), self.n, eps, decay, _stream()),
                   "ds_bn_finalize_multi")


def bn_bwd_finalize_multi(sum_segs, M, C_, betas, dbetas, coef):
    """ds_bn_bwd_finalize_multi: segment i of sum_segs is a LAYER with its own beta / dbeta vector (dbetas[i] may be None)."""
    n = sum_segs.nseg
    b = (C.c_void_p * 4)(*[t.data_ptr() for t in betas[:n]])
    d = (C.c_void_p * 4)(*[(t.data_ptr() if t is not None else None) for t in dbetas[:n]])
    _lib.check(_lib.load().ds_bn_bwd_finalize_multi(C.byref(sum_segs), M, C_, C.cast(b, C.c_void_p), C.cast(d, C.c_void_p),
                                                    _p(coef), _stream()), "ds_bn_bwd_finalize_multi")


def bn_finalize_apply_relu(stats, P, count, C_, beta, eps, decay, mean, rstd, shift, mm, mv, pivot, z, M, segs, ticket):
    """ds_bn_finalize + ds_bn_apply_relu as one launch (ticket: two zero-initialised int32 words of the layer)."""
    _lib.check(_lib.load().ds_bn_finalize_apply_relu(_p(stats), P, count, C_, _p(beta), _p(pivot), eps, decay, _p(mean), _p(rstd),
                                                     _p(shift), _p(mm), _p(mv), _p(z), M, C.byref(segs), _p(ticket), _stream()),
               "ds_bn_finalize_apply_relu")


def bn_bwd_finalize_apply(sum_segs, M, C_, beta, dbeta, coef, z, segs, mean, rstd, shift, dz, ticket, amax=None, ldz=0,
                          betas=None, dbetas=None):
    """ds_bn_bwd_finalize_segs (beta / dbeta: one vector) or ds_bn_bwd_finalize_multi (betas / dbetas: the segments' own) and
    the ds_bn_bwd_apply pass behind it as one launch.  dz: fp32 over z's layout, or a dense bf16 tensor."""
    bv = dv = None
    if betas is not None:
        n = sum_segs.nseg
        bv = C.cast((C.c_void_p * 4)(*[t.data_ptr() for t in betas[:n]]), C.c_void_p)
        dv = C.cast((C.c_void_p * 4)(*[(t.data_ptr() if t is not None else None) for t in dbetas[:n]]), C.c_void_p)
    out16 = dz.dtype == torch.bfloat16
    _lib.check(_lib.load().ds_bn_bwd_finalize_apply(C.byref(sum_segs), _p(beta), _p(dbeta), bv, dv, _p(coef), _p(z), ldz or C_,
                                                    C.byref(segs), M, C_, _p(mean), _p(rstd), _p(shift), _p(dz),
                                                    DS_DTYPE_BF16 if out16 else DS_DTYPE_F32, dz.stride(0) if out16 else 0,
                                                    _p(amax), _p(ticket), _stream()), "ds_bn_bwd_finalize_apply")


def bn_bwd_finalize(partials, P, M, C_, dbeta, coef):
    _lib.check(_lib.load().ds_bn_bwd_finalize(_p(partials), P, M, C_, _p(dbeta), _p(coef), _stream()),
               "ds_bn_bwd_finalize")


def bn_bwd_apply(z, segs, M, C_, mean, rstd, shift, coef, dz, amax=None, ldz=0):
    """dz: fp32 (over z, or any tensor with z's row stride), or a SEPARATE dense bf16 tensor [M, C] -- the form the 16-bit
    configurations' 1x1 input gradients read (ds_bn_bwd_apply_bf16)."""
    if z.dtype == torch.bfloat16:          # z in bf16 storage: dz into its own bf16 tensor
        assert dz.dtype == torch.bfloat16
        _lib.check(_lib.load().ds_bn_bwd_apply_z16(_p(z), ldz or C_, C.byref(segs), M, C_, _p(mean), _p(rstd), _p(shift),
                                                   _p(coef), _p(dz), dz.stride(0), _p(amax), _stream()), "ds_bn_bwd_apply_z16")
        return
    if dz.dtype == torch.bfloat16:
        _lib.check(_lib.load().ds_bn_bwd_apply_bf16(_p(z), ldz or C_, C.byref(segs), M, C_, _p(mean), _p(rstd), _p(shift),
                                                    _p(coef), _p(dz), dz.stride(0), _p(amax), _stream()), "ds_bn_bwd_apply_bf16")
        return
    _lib.check(_lib.load().ds_bn_bwd_apply(_p(z), ldz or C_, C.byref(segs), M, C_, _p(mean), _p(rstd), _p(shift), _p(coef),
                                           _p(dz), _p(amax), _stream()), "ds_bn_bwd_apply")


def maxpool_fwd(x, y, argmax, N, H, W, C_, k, stride, mode="SAME"):
    if mode == "SAME":
        OH, pt = same_pad(H, k, stride)
        OW, pl = same_pad(W, k, stride)
    else:
        OH, OW, pt, pl = (H - k) // stride + 1, (W - k) // stride + 1, 0, 0
    assert x.dtype == y.dtype
    _lib.check(_lib.load().ds_maxpool_fwd(_p(x), _p(y), _p(argmax), N, H, W, C_, k, stride, pt, pl, OH, OW,
                                          act_dtype(x), _stream()), "ds_maxpool_fwd")
    return OH, OW


def maxpool_bn_relu_fwd(z, rstd, shift, y, argmax, N, H, W, C_, k, stride, amax=None):
    """y = maxpool(relu(z*rstd + shift)) computed as relu(rstd*maxpool(z) + shift): SAME padding, 3x3 only."""
    OH, pt = same_pad(H, k, stride)
    OW, pl = same_pad(W, k, stride)
    _lib.check(_lib.load().ds_maxpool_bn_relu_fwd(_p(z), _p(rstd), _p(shift), _p(y), _p(argmax), N, H, W, C_, k, stride,
                                                  pt, pl, OH, OW, act_dtype(y), _p(amax), _stream()),
               "ds_maxpool_bn_relu_fwd")
    return OH, OW


def bn_pool_bwd_partials(N, H, W, C_):
    OH, _ = same_pad(H, 3, 2)
    OW, _ = same_pad(W, 3, 2)
    return _lib.load().ds_bn_pool_bwd_partials(N, OH, OW, C_)


def bn_pool_bwd_reduce(z, dpool, argmax, N, H, W, C_, mean, rstd, shift, partials):
    """BatchNorm(+ReLU) backward sums of a conv behind a 3x3/2 SAME max pool, from the pooled gradient."""
    OH, pt = same_pad(H, 3, 2)
    OW, pl = same_pad(W, 3, 2)
    _lib.check(_lib.load().ds_bn_pool_bwd_reduce(_p(z), _p(dpool), _p(argmax), N, H, W, C_, pt, pl, OH, OW, _p(mean),
                                                 _p(rstd), _p(shift), _p(partials), _stream()), "ds_bn_pool_bwd_reduce")


def bn_pool_bwd_apply(z, dpool, argmax, N, H, W, C_, mean, rstd, shift, coef, dz):
    OH, pt = same_pad(H, 3, 2)
    OW, pl = same_pad(W, 3, 2)
    _lib.check(_lib.load().ds_bn_pool_bwd_apply(_p(z), _p(dpool), _p(argmax), N, H, W, C_, pt, pl, OH, OW, _p(mean),
                                                _p(rstd), _p(shift), _p(coef), _p(dz), _stream()), "ds_bn_pool_bwd_apply")


def maxpool_bwd(dy, argmax, dx, accumulate, N, H, W, C_, k, stride, mode="SAME"):
    if dy.dtype == torch.bfloat16:          # (3x3 / 1 SAME only: Branch_3's pool behind a dgrad that writes bf16)
        assert k == 3 and stride == 1 and mode == "SAME"
        _lib.check(_lib.load().ds_maxpool3_bwd_dy16(_p(dy), _p(argmax), _p(dx), int(accumulate), None, DS_DTYPE_F32, N, H, W, C_,
                                                    None, _stream()), "ds_maxpool3_bwd_dy16")
        return
    if mode == "SAME":
        OH, pt = same_pad(H, k, stride)
        OW, pl = same_pad(W, k, stride)
    else:
        OH, OW, pt, pl = (H - k) // stride + 1, (W - k) // stride + 1, 0, 0
    _lib.check(_lib.load().ds_maxpool_bwd(_p(dy), _p(argmax), _p(dx), int(accumulate), N, H, W, C_, k, stride, pt,
                                          pl, OH, OW, _stream()), "ds_maxpool_bwd")


def maxpool3_bwd_sums_partials(N, W, C_):
    return _lib.load().ds_maxpool3_bwd_sums_partials(N, W, C_)


def maxpool3_bwd_sums(dy, argmax, dx, accumulate, y, N, H, W, C_, partials):
    """ds_maxpool_bwd of a 3x3 / 1 SAME pool + the BatchNorm-backward sums (sum g, sum g*y over y > 0) of the activation y."""
    if dy.dtype == torch.bfloat16:          # the pool's output gradient in bf16 storage
        _lib.check(_lib.load().ds_maxpool3_bwd_dy16(_p(dy), _p(argmax), _p(dx), int(accumulate), _p(y), act_dtype(y), N, H, W, C_,
                                                    _p(partials), _stream()), "ds_maxpool3_bwd_dy16")
        return
    _lib.check(_lib.load().ds_maxpool3_bwd_sums(_p(dy), _p(argmax), _p(dx), int(accumulate), _p(y), act_dtype(y), N, H, W, C_,
                                                _p(partials), _stream()), "ds_maxpool3_bwd_sums")


def avgpool_dropout_fwd(x, N, HW, C_, keep, seed, mask_in, mask_out, out, seed_dev=None):
    _lib.check(_lib.load().ds_avgpool_dropout_fwd(_p(x), N, HW, C_, keep, seed, _p(seed_dev), _p(mask_in),
                                                  _p(mask_out), _p(out), _stream()), "ds_avgpool_dropout_fwd")


def avgpool_dropout_bwd(dout, mask, N, HW, C_, keep, dx):
    _lib.check(_lib.load().ds_avgpool_dropout_bwd(_p(dout), _p(mask), N, HW, C_, keep, _p(dx), _stream()),
               "ds_avgpool_dropout_bwd")


def gather_rows(table, ids, out, B, T, D, time_major=True):
    _lib.check(_lib.load().ds_gather_rows(_p(table), _p(ids), _p(out), B, T, D, table.shape[0], int(time_major),
                                          _stream()), "ds_gather_rows")


def embedding_grad(dx, ids, dtable, B, T, D, time_major=True):
    _lib.check(_lib.load().ds_embedding_grad(_p(dx), _p(ids), _p(dtable), B, T, D, dtable.shape[0], int(time_major),
                                             _stream()), "ds_embedding_grad")


def lstm_cell_fwd(gates, c_prev, h_prev, seq_len, t, B, H, forget_bias, c_out, h_out, rec_slabs=None, nslabs=0,
                  slab_stride=0):
    _lib.check(_lib.load().ds_lstm_cell_fwd(_p(gates), _p(rec_slabs), nslabs, slab_stride, _p(c_prev), _p(h_prev),
                                            _p(seq_len), t, B, H, forget_bias, _p(c_out), _p(h_out), _stream()),
               "ds_lstm_cell_fwd")


def lstm_cell_bwd(acts, c_t, c_prev, dh, dc, seq_len, t, B, H, dgates, dc_prev, dh_carry, dh_slabs=None, nslabs=0,
                  slab_stride=0):
    _lib.check(_lib.load().ds_lstm_cell_bwd(_p(acts), _p(c_t), _p(c_prev), _p(dh), _p(dh_slabs), nslabs, slab_stride,
                                            _p(dc), _p(seq_len), t, B, H, _p(dgates), _p(dc_prev), _p(dh_carry),
                                            _stream()), "ds_lstm_cell_bwd")


def lstm_seq_supported(B, H):
    return bool(_lib.load().ds_lstm_seq_supported(B, H))


def lstm_seq_workspace(B, H):
    return int(_lib.load().ds_lstm_seq_workspace(B, H))


def lstm_seq_fwd(gates, wh_ptr, ldw, h, c, seq_len, T, B, H, forget_bias, ws, rows=1):
    """rows: row groups per workgroup (1, 2, 4, 8) -- scheduling only.  `ws` is zeroed once by its owner."""
    _lib.check(_lib.load().ds_lstm_seq_fwd(_p(gates), wh_ptr, ldw, _p(h), _p(c), _p(seq_len), T, B, H, forget_bias,
                                           int(rows), _p(ws), ws.numel() * ws.element_size(), _stream()),
               "ds_lstm_seq_fwd")


def lstm_seq_bwd(acts, wh_ptr, ldw, c, dh_last, ld_dh, seq_len, T, B, H, dgates, ws, rows=1):
    _lib.check(_lib.load().ds_lstm_seq_bwd(_p(acts), wh_ptr, ldw, _p(c), _p(dh_last), ld_dh, _p(seq_len), T, B, H,
                                           _p(dgates), int(rows), _p(ws), ws.numel() * ws.element_size(), _stream()),
               "ds_lstm_seq_bwd")


def seq_sort_desc(seq_len, B, T, perm, len_sorted):
    """perm [B] int32, len_sorted [B] int64: the batch in descending order of length (ties by index), on the device."""
    _lib.check(_lib.load().ds_seq_sort_desc(_p(seq_len), B, T, _p(perm), _p(len_sorted), _stream()), "ds_seq_sort_desc")


def permute_rows(src, dst, perm, rows, cols, gather=True):
    """gather: dst[j] = src[perm[j]]; scatter (gather=False): dst[perm[j]] = src[j].  fp32 or int64 rows."""
    assert src.element_size() == dst.element_size() and src.element_size() in (4, 8)
    _lib.check(_lib.load().ds_permute_rows(_p(src), src.stride(0), _p(dst), dst.stride(0), _p(perm), rows, cols, src.element_size(),
                                           int(bool(gather)), _stream()), "ds_permute_rows")


def lstm_seq_status(ws, B):
    """0 = ok; call after a synchronise (it copies two words to the host).  Raises on a hand-off timeout of any
    forward (bit 0) or backward (bit 1) launch since the last read: the error words are sticky until reported, then cleared."""
    rc = _lib.load().ds_lstm_seq_status(_p(ws), B)
    if rc != 0:
        which = " and ".join(n for b, n in ((1, "forward"), (2, "backward")) if rc > 0 and rc & b) or "status read"
        raise RuntimeError("ds_lstm_seq: a workgroup hand-off of the %s launch timed out (status %d): the LSTM results "
                           "of this step are invalid (a workgroup of a row group never became resident)" % (which, rc))
    return rc


def softmax_ce(logits, labels, B, C_, grad_scale, grad_scale_dev, loss, dlogits):
    _lib.check(_lib.load().ds_softmax_ce(_p(logits), _p(labels), B, C_, grad_scale, _p(grad_scale_dev), _p(loss),
                                         _p(dlogits), _stream()), "ds_softmax_ce")


def adam_tf(theta, g, m, v, n, n_wd, wd, grad_scale, lr_t, b1, b2, eps, lr_t_dev=None):
    _lib.check(_lib.load().ds_adam_tf(_p(theta), _p(g), _p(m), _p(v), n, n_wd, wd, grad_scale, lr_t, _p(lr_t_dev),
                                      b1, b2, eps, _stream()), "ds_adam_tf")


def sumsq(x, n, scratch, out):
    _lib.check(_lib.load().ds_sumsq(_p(x), n, _p(scratch), _p(out), _stream()), "ds_sumsq")


def colsum(x, M, C_, ld, scratch, out):
    _lib.check(_lib.load().ds_colsum(_p(x), M, C_, ld, _p(scratch), _p(out), _stream()), "ds_colsum")


def copy2d(src, lds, dst, ldd, rows, cols):
    _lib.check(_lib.load().ds_copy2d(_p(src), lds, _p(dst), ldd, rows, cols, _stream()), "ds_copy2d")


def pad_channels(src, cs, dst, cd, pixels):
    _lib.check(_lib.load().ds_pad_channels(_p(src), cs, _p(dst), cd, pixels, _stream()), "ds_pad_channels")


def fill(dst, n, value):
    _lib.check(_lib.load().ds_fill(_p(dst), n, value, _stream()), "ds_fill")
