"""Inception-v1 image tower: forward + backward orchestration of the HIP kernels.

What is computed is fixed by the reference (image_model/inception_v1.py:29-309 under
slim/nets/inception_utils.py:32-71); how is MI355X-first:

  * NHWC activations; every conv is one implicit-GEMM MFMA launch reading the TF HWIO weights in
    place; the three 1x1 convs that read a block's input run as ONE GEMM (N = b0+b1a+b2a);
  * BatchNorm statistics come out of the conv epilogue; normalise+ReLU writes the branch outputs
    straight into channel slices of the block's concat buffer (tf.concat never materialises);
  * backward: BN/ReLU backward overwrites the saved pre-activation z with dz in place; dgrad is the
    same MFMA kernel with flipped taps over the untouched HWIO weights; wgrad only for the
    trainable scope (Mixed_5c :229-250 and Logits :302-303); BatchNorm beta gradients for all 57
    layers;
  * all buffers are allocated once per batch size; nothing is allocated or synchronised per step.
"""
import ctypes as C
import os

import torch

from . import _lib, ops
from .ops import WgradPlan, gemm_plan, make_segments, same_pad, DS_EPI_BIAS, DS_EPI_STATS

BN_EPS = 0.001         # slim/nets/inception_utils.py:35
BN_DECAY = 0.9997      # slim/nets/inception_utils.py:34
WEIGHT_DECAY = 0.00004  # slim/nets/inception_utils.py:32

# Topology of image_model/inception_v1.py (endpoint names and channel counts, :62-250)
TOPOLOGY = [
    ("conv", "Conv2d_1a_7x7", 7, 2, 64),
    ("maxpool", "MaxPool_2a_3x3", 3, 2),
    ("conv", "Conv2d_2b_1x1", 1, 1, 64),
    ("conv", "Conv2d_2c_3x3", 3, 1, 192),
    ("maxpool", "MaxPool_3a_3x3", 3, 2),
    ("mixed", "Mixed_3b", 64, (96, 128), (16, 32), 32),
    ("mixed", "Mixed_3c", 128, (128, 192), (32, 96), 64),
    ("maxpool", "MaxPool_4a_3x3", 3, 2),
    ("mixed", "Mixed_4b", 192, (96, 208), (16, 48), 64),
    ("mixed", "Mixed_4c", 160, (112, 224), (24, 64), 64),
    ("mixed", "Mixed_4d", 128, (128, 256), (24, 64), 64),
    ("mixed", "Mixed_4e", 112, (144, 288), (32, 64), 64),
    ("mixed", "Mixed_4f", 256, (160, 320), (32, 128), 128),
    ("maxpool", "MaxPool_5a_2x2", 2, 2),
    ("mixed", "Mixed_5b", 256, (160, 320), (32, 128), 128),
    ("mixed", "Mixed_5c", 384, (192, 384), (48, 128), 128),
]
ENDPOINTS = [t[1] for t in TOPOLOGY]
AMAX_RECORDS = 160     # fp8: max|.| records handed out by InceptionV1Engine.new_amax (about 91 in use)
TRAINABLE_ENDPOINTS = ("Mixed_5c",)      # inception_v1.py:229-231; earlier scopes are trainable=False (:57-59)


def _vp(addr):
    return C.c_void_p(addr)


class ConvBN:
    """slim.conv2d under inception_arg_scope: conv (no bias) -> BatchNorm(train, beta only) -> ReLU.
    `scopes` lists (tf_scope, c0, c1): more than one entry = horizontally fused 1x1 convs."""

    def __init__(self, eng, scopes, k, stride, cin, cout, H, W, trainable, beta_bucket, fold=False):
        self.eng, self.scopes, self.k, self.stride = eng, scopes, k, stride
        self.cin, self.cout, self.H, self.W = cin, cout, H, W
        self.trainable, self.fold = trainable, fold
        self.pool_inside = False   # Conv2d_1a_7x7 only: MaxPool_2a inside the conv kernel (z = the window maxima)
        self.slot = 0          # which of the engine's scratch sets (one per branch stream) this layer uses
        self.OH, _ = same_pad(H, k, stride)
        self.OW, _ = same_pad(W, k, stride)
        st = eng.store
        self.key = scopes[0][0] if len(scopes) == 1 else scopes[0][0].rsplit("/", 2)[0] + "/fused_1x1"
        cin_store = 4 if fold else cin            # stem: Cin 3 zero-padded to 4 (one float4 per tap)
        cols = lambda suffix: [(s + suffix, c0, c1) for (s, c0, c1) in scopes]
        st.declare(self.key + "/weights", (k, k, cin_store, cout), trainable, l2=trainable, bucket=1,
                   columns=cols("/weights"))
        st.declare(self.key + "/BatchNorm/beta", (cout,), eng.trainable_bn_beta, bucket=beta_bucket,
                   columns=cols("/BatchNorm/beta"))
        st.declare(self.key + "/BatchNorm/moving_mean", (cout,), False, columns=cols("/BatchNorm/moving_mean"))
        st.declare(self.key + "/BatchNorm/moving_variance", (cout,), False,
                   columns=cols("/BatchNorm/moving_variance"))

    def alloc(self, B):
        eng, dev = self.eng, self.eng.device
        k, cin, cout = self.k, self.cin, self.cout
        self.B = B
        self.M = B * self.OH * self.OW
        self.ldz = cout               # row stride of z (use_concat_slice: the conv writes into the block's concat buffer)
        self.skip_apply = False       # ... and its consumers apply BatchNorm + ReLU on load
        self.mean = torch.empty(cout, device=dev)
        self.rstd = torch.empty(cout, device=dev)
        self.shift = torch.empty(cout, device=dev)
        self.coef = torch.empty(2, cout, device=dev)
        self.fa_ticket = torch.zeros(4, dtype=torch.int32, device=dev)      # finalize + apply as one launch: [0:2] forward, [2:4] backward
        self._plain_segs = None
        # The library plans the launch (ds_conv_plan): kernel family per shape and arithmetic -- implicit GEMM / wide 1x1,
        # fused Winograd F(2x2) / F(4x4) by its launch-time model, the packed-RGB stem kernel, the register-direct bf16 /
        # fp8 / f32x3 kernels -- the BatchNorm partial count and the prepared filter form.  The engine only says what the
        # layer is (one conv entry point; the selection rules live next to the kernels).
        opts = eng.plan_options()
        if self.fold:      # Conv2d_1a_7x7 reads the packed RGB batch (or, generic kernel, its zero-padded 4-channel copy)
            if self.pool_inside:          # ... and takes MaxPool_2a inside the kernel (ConvStage.alloc decides)
                opts |= ops.DS_PLAN_STEM_POOL
            self.fwd = ops.LayerPlan(ops.DS_CONV_FWD, eng.arith, opts | ops.DS_PLAN_PACKED_RGB, B, self.H, self.W, 4, cout,
                                     k, self.stride, 4, cout, DS_EPI_STATS)
            self.pool_inside = self.fwd.family == ops.DS_FAM_STEM_POOL
        else:
            self.fwd = ops.LayerPlan(ops.DS_CONV_FWD, eng.arith, opts, B, self.H, self.W, cin, cout, k, self.stride, 0, cout,
                                     DS_EPI_STATS)
        self.fwd.alloc_weights(dev)
        # z: the pre-BatchNorm conv output, kept for the backward pass.  pool_inside: the 3x3 / 2 window maxima of it
        # [B, OH/2, OW/2, cout] -- the only thing the fused stem writes
        self.z = torch.empty(B * (self.OH // 2) * (self.OW // 2) if self.pool_inside else self.M, cout, device=dev)
        self.stem_direct = self.fwd.family in (ops.DS_FAM_STEM, ops.DS_FAM_STEM_POOL)
        # fp8: device records with max|x| of the forward input / of dz for the layers ds_conv_fp8 can take
        fp8 = eng.dtype == "fp8" and not self.fold and k in (1, 3) and self.stride == 1
        if fp8:
            self.amax = torch.zeros(2, ops.AMAX_FLOATS, device=dev)      # fallback records for ds_absmax passes
            self.dz_amax = eng.new_amax()                # max|dz|, collected by ds_bn_bwd_apply
        self.u_version = -1
        eng.need_stats(self.fwd.partials * 2 * cout)
        self.bwd_P = ops.bn_bwd_partials(self.M, cout)
        eng.need_bwd_partials(self.bwd_P * 2 * cout)
        self.pool_P = ops.bn_pool_bwd_partials(B, self.OH, self.OW, cout)      # used when a 3x3/2 pool follows
        eng.need_bwd_partials(self.pool_P * 2 * cout)
        self.dgrad = None
        self.z16 = False              # z in bf16 storage (make_dgrad)
        self.wgrad = None
        self.dy_parts = None          # set_dy_parts(): where the gradient of this layer's output lives
        self._dz_amax_live = False
        self.dx_sums = None           # this layer's dgrad emits the consumer's BatchNorm sums (emit_dx_sums)
        self.bnb = False              # the dgrad forms dz from z and dy as it loads (enable_bnb): no ds_bn_bwd_apply pass
        self.dx_y = None
        self._sum_segs = None
        if self.trainable:
            if self.fold:      # stem (train_all only): KW folded into the channel axis like the forward conv
                self.wgrad = WgradPlan(B, self.H, self.W, 7 * 4, 4, 7, 1, self.stride, cout, cout, fold_cin=4)
            else:
                self.wgrad = WgradPlan(B, self.H, self.W, cin, 0, k, k, self.stride, cout, cout)
            eng.need_ws(self.wgrad.ws_bytes)

    def use_concat_slice(self, zview, ld, rstd, shift, mean=None):
        """The conv output goes straight into this layer's channel slice of the block's concat buffer (row stride ld)
        and stays PRE-BatchNorm there: the consumers of the concat apply relu(z*rstd + shift) as they load it
        (MixedStage zcat), so the layer's BatchNorm-apply pass and its dense z buffer disappear.  rstd / shift: this
        layer's slices of the block's per-channel arrays."""
        self.z, self.ldz, self.rstd, self.shift, self.skip_apply = zview, ld, rstd, shift, True
        if mean is not None:          # (MixedStage batch_bn: one ds_bn_bwd_apply over the three block-closing layers' columns)
            self.mean = mean
        self.fwd.d.ldz = ld

    @property
    def zmean(self):
        """mean / shift as the passes over THIS layer's z take them (z16: z is stored centred about the pivot)."""
        return self.mean_c if self.z16 else self.mean

    @property
    def zshift(self):
        return self.shift_c if self.z16 else self.shift

    def plan_finalize(self):
        """ds_bn_finalize inside the forward conv launch where the library can (wide 1x1 kernel, at most 256 partials: the
        small per-GPU batches, where a dependent launch costs more than its work).  Called once the plan is final
        (MixedStage.alloc may still switch Branch_3's conv to the pooling loader)."""
        eng = self.eng
        self.fin = None
        n = self.fwd.finalize_tickets() if eng.fuse_finalize else 0
        if n > 0:
            self.fin_tickets = torch.zeros(n, dtype=torch.int32, device=eng.device)
            f = _lib.BnFinalizeInLaunch()
            f.ticket, f.eps, f.decay = self.fin_tickets.data_ptr(), BN_EPS, BN_DECAY
            self.fin = f

    def bind(self):
        st = self.eng.store
        self.w_ptr = _vp(st.ptr(self.key + "/weights"))
        self.beta = st.view(self.key + "/BatchNorm/beta")
        self.mm = st.view(self.key + "/BatchNorm/moving_mean")
        self.mv = st.view(self.key + "/BatchNorm/moving_variance")
        self.mean.copy_(self.mm)          # first pivot of the batch statistics (ConvBN.forward)
        eng = self.eng
        self.stats_buf, self.bwdp_buf, self.ws_buf = eng.stats_set[self.slot], eng.bwdp_set[self.slot], eng.ws_set[self.slot]
        for plan in (self.fwd, self.dgrad):          # split-K Winograd launches: the slices' partial outputs
            if plan is not None and plan.ws_bytes:
                plan.set_workspace(eng.cws_set[self.slot])
        self.plan_finalize()
        if self.fin is not None:
            f = self.fin
            f.beta, f.mean, f.rstd, f.shift = self.beta.data_ptr(), self.mean.data_ptr(), self.rstd.data_ptr(), self.shift.data_ptr()
        self.gw_ptr = _vp(st.grad_ptr(self.key + "/weights")) if self.trainable else None
        self.gbeta = st.grad_view(self.key + "/BatchNorm/beta") if self.eng.trainable_bn_beta else None

    def set_dy_parts(self, parts):
        """parts: [(c0, c1, address, ld)] -- the channel ranges of this layer's output gradient and where each lives
        (slices of the block's concat gradient, the reduce buffers' gradients).  part_sums[i] is filled in by the
        stage whose dgrad WRITES that part when it can also emit the part's BatchNorm sums (DS_EPI_BNSUMS):
        (partials tensor, P, first column of the part in the producer's output, producer's column count)."""
        self.dy_parts = parts
        self.dy_segs = make_segments(parts)
        self.part_segs = [make_segments([(0, c1 - c0, ptr, ld)]) for (c0, c1, ptr, ld) in parts]
        self.part_sums = [None] * len(parts)
        self.part_sums2 = [None] * len(parts)    # a second source of the same form: the part's gradient lives in TWO tensors (dy2)
        self.dy2 = False                         # some part has a second addend (dy_segs.ptr2): see MixedStage.alloc, split_dout
        self.part_pool = [None] * len(parts)     # (pool stage, first column): the part feeds nothing but that max pool
        self._sum_segs = None

    def emit_dx_sums(self, y):
        """This layer's dgrad writes the gradient of `y` (an activation relu(bn(.)) with the dgrad output's pixel
        stride): have its epilogue emit the column sums that layer's BatchNorm backward needs.  Returns
        (partials tensor, P) or None when the dgrad kernel of this shape cannot (implicit-GEMM fallbacks, bf16)."""
        eng = self.eng
        if not eng.bwd_sums or self.dgrad is None:
            return None
        P = self.dgrad.enable_bnsums(self.dgrad.d.ldz)
        if not P:
            return None
        self.dgrad.d.mask_dtype = ops.act_dtype(y)          # (bf16 under 16-bit activation storage)
        self.dx_sums = torch.empty(2 * self.cin * P, device=eng.device)
        self.dx_y = y
        return self.dx_sums, P

    def _sum_plan(self):
        """Where sum g and sum g*xhat of this layer come from: parts whose producer emitted them are taken as they are, the
        others are reduced over their column range (_run_reduce_jobs).  Built once per allocation."""
        eng = self.eng
        M, Cc = self.M, self.cout
        if self._sum_segs is None:
            sg = ops.SumSegments()
            sg.nseg = len(self.dy_parts)
            self._reduce_jobs = []
            self._sync_views = []          # sync_bn: the partial-sum regions to all-reduce before the finalize
            scratch, P0 = self.bwdp_buf.data_ptr(), self.bwd_P
            for i, (c0, c1, _, _) in enumerate(self.dy_parts):
                sg.c_begin[i], sg.c_end[i] = c0, c1
                src = self.part_sums[i]
                if src is not None:
                    buf, P, off, ctot = src
                    sg.P[i], sg.kind[i] = P, 1
                    sg.s[i] = buf.data_ptr() + 4 * off * P
                    sg.q[i] = buf.data_ptr() + 4 * (ctot + off) * P
                    n = c1 - c0
                    self._sync_views += [buf[off * P:(off + n) * P], buf[(ctot + off) * P:(ctot + off + n) * P]]
                    if self.part_sums2[i] is not None:          # the other addend's sums, added by the finalize
                        buf2, P2, off2, ctot2 = self.part_sums2[i]
                        sg.P2[i] = P2
                        sg.s2[i] = buf2.data_ptr() + 4 * off2 * P2
                        sg.q2[i] = buf2.data_ptr() + 4 * (ctot2 + off2) * P2
                        self._sync_views += [buf2[off2 * P2:(off2 + n) * P2], buf2[(ctot2 + off2) * P2:(ctot2 + off2 + n) * P2]]
                elif self.part_pool[i] is not None:
                    # The part feeds only a max pool.  Every window hands its gradient to ONE input pixel p*, whose
                    # activation is the pooled value, so   sum_pixels g = sum_windows dpool (ypool > 0)   and
                    # sum_pixels g*xhat = sum_windows dpool (ypool - beta) (ypool > 0)   -- both sums from the POOLED
                    # tensors (a quarter of the elements; z is not read at all).  That is ds_bn_bwd_reduce run on
                    # (z := ypool, mean := beta, rstd := 1, shift := 0).
                    pool, off = self.part_pool[i]
                    n = c1 - c0
                    Mp = pool.B * pool.H * pool.W
                    Pp = ops.bn_bwd_partials(Mp, n)
                    sg.P[i], sg.kind[i] = Pp, 0
                    sg.s[i], sg.q[i] = scratch, scratch + 4 * n * Pp
                    o0 = (scratch - self.bwdp_buf.data_ptr()) // 4
                    self._sync_views.append(self.bwdp_buf[o0:o0 + 2 * n * Pp])
                    seg = make_segments([(0, n, pool.dout.data_ptr() + 4 * off, pool.C)])
                    if getattr(pool, "raw", False):
                        # the pool's output holds the window maxima of z itself (pool_inside): the plain reduce on
                        # (zmax, mean, rstd, shift) -- the same predicate rstd * zmax + shift > 0, the same sum of g
                        stat = (_vp(self.mean.data_ptr() + 4 * c0), _vp(self.rstd.data_ptr() + 4 * c0), _vp(self.shift.data_ptr() + 4 * c0))
                    else:
                        stat = (_vp(self.beta.data_ptr() + 4 * c0), ops._p(eng.ones), ops._p(eng.zeros))
                    self._reduce_jobs.append(("pool", seg, Mp, n, _vp(pool.out.data_ptr() + pool.out.element_size() * off),
                                              pool.C, stat, _vp(scratch), ops.act_dtype(pool.out)))
                    scratch += 4 * 2 * n * Pp
                else:
                    n = c1 - c0
                    sg.P[i], sg.kind[i] = P0, 0
                    sg.s[i], sg.q[i] = scratch, scratch + 4 * n * P0
                    o0 = (scratch - self.bwdp_buf.data_ptr()) // 4
                    self._sync_views.append(self.bwdp_buf[o0:o0 + 2 * n * P0])
                    self._reduce_jobs.append(("full", i, c0, n, _vp(scratch)))
                    scratch += 4 * 2 * n * P0
            self._sum_segs = sg
        return self._sum_segs

    def _run_reduce_jobs(self):
        eng = self.eng
        M = self.M
        for job in self._reduce_jobs:
            if job[0] == "pool":
                _, seg, Mp, n, yp, ldy, stat, dst, ydt = job
                ops.bn_bwd_reduce(yp, seg, Mp, n, stat[0], stat[1], stat[2], dst, ldz=ldy, z_dtype=ydt)
                continue
            _, i, c0, n, dst = job
            off = 4 * c0
            ops.bn_bwd_reduce(_vp(self.z.data_ptr() + self.z.element_size() * c0), self.part_segs[i], M, n, _vp(self.zmean.data_ptr() + off),
                              _vp(self.rstd.data_ptr() + off), _vp(self.zshift.data_ptr() + off), dst, ldz=self.ldz,
                              z_dtype=ops.act_dtype(self.z))

    def _bn_bwd_sums(self):
        """Sum g and sum g*xhat of this layer (_sum_plan, _run_reduce_jobs) and one finalize launch."""
        eng = self.eng
        M, Cc = self.M, self.cout
        self._sum_plan()
        self._run_reduce_jobs()
        if eng.sync_bn:
            # beta's gradient stays this rank's own sum (the gradient all-reduce adds the ranks); the two column MEANS of the
            # backward formula are over the global batch: finalize once locally for dbeta, all-reduce, finalize again
            if self.gbeta is not None:
                ops.bn_bwd_finalize_segs(self._sum_segs, M, Cc, self.beta, self.gbeta, self.coef)
            for v in self._sync_views:
                eng.all_reduce(v)
            ops.bn_bwd_finalize_segs(self._sum_segs, M * eng.sync_world, Cc, self.beta, None, self.coef)
            return
        ops.bn_bwd_finalize_segs(self._sum_segs, M, Cc, self.beta, self.gbeta, self.coef)

    def _finalize_plain(self, P):
        """ds_bn_bwd_finalize of bwdp_buf [2][C][P] (sync_bn: as in _bn_bwd_sums)."""
        eng = self.eng
        M, Cc = self.M, self.cout
        gb = self.gbeta if self.gbeta is not None else eng.dummy
        if not eng.sync_bn:
            ops.bn_bwd_finalize(self.bwdp_buf, P, M, Cc, gb, self.coef)
            return
        if self.gbeta is not None:
            ops.bn_bwd_finalize(self.bwdp_buf, P, M, Cc, gb, self.coef)
        eng.all_reduce(self.bwdp_buf[:2 * Cc * P])
        ops.bn_bwd_finalize(self.bwdp_buf, P, M * eng.sync_world, Cc, eng.dummy, self.coef)

    def make_dgrad(self, lddx, allow_z16=False):
        """Conv2DBackpropInput as a forward conv over dz with flipped taps (stride-1 SAME convs only); the library picks
        the kernel family for the swapped shape.  allow_z16: the caller's forward / backward use of this layer's z goes through
        ds_bn_apply_relu / ds_bn_bwd_reduce / ds_bn_bwd_apply only (the plain layers of a Mixed block), so z may live in bf16."""
        assert self.stride == 1
        eng = self.eng
        self.dgrad = ops.LayerPlan(ops.DS_CONV_DGRAD, eng.arith, eng.plan_options(), self.B, self.H, self.W, self.cin,
                                   self.cout, self.k, 1, self.ldz, lddx, 0)
        self.dgrad.alloc_weights(eng.device)
        # BatchNorm + ReLU backward APPLIED ON LOAD by the wide 1x1 dgrad (ds_conv_desc.bnb): the layer skips its
        # ds_bn_bwd_apply pass and the dgrad reads z and the activation gradient instead of dz (frozen layers only: a weight
        # gradient needs dz).  Bit-identical, but it pays only where the dgrad is HBM-bound with ONE column tile -- Conv2d_2b
        # (219 -> 190 us): the second A stream and ~8 VALU per element are redone for every column tile and cost the
        # matrix-bound dgrads 25-65 % (profiles/r04_bnb_layers.txt: all fifteen 1x1 shapes; the whole step 15.2 -> 16.0 ms
        # with it everywhere).  bnb_on_load: 1 = where it wins (default), 2 = every frozen 1x1 layer (tests), 0 = off
        # 16-bit configurations: dz of a frozen layer goes to a SEPARATE bf16 tensor (ds_bn_bwd_apply_bf16) when its dgrad runs on
        # the register-direct bf16 kernel (1x1) or on F(4x4) with bf16 pieces (3x3), both of which round dz to bf16 as they load
        # anyway: same bits, 2 B less written and 2 B less read per element of every gradient stream of the frozen tower
        self.dz16 = None
        fam_ok = (self.k == 1 and self.dgrad.family == ops.DS_FAM_BF16D and self.cout % 8 == 0) or \
                 (self.k == 3 and self.dgrad.family == ops.DS_FAM_WINO4H and eng.dz16 >= 2)
        if (eng.dz16 and eng.act16 and not self.trainable and fam_ok and self.dgrad.x16_ok and self.ldz == self.cout):
            self.dz16 = torch.empty(self.M, self.cout, device=eng.device, dtype=torch.bfloat16)
            self.dgrad.d.ldx = self.cout
        # z16: z of such a layer itself in bf16 storage (ds_conv_desc.z_dtype; statistics from the fp32 accumulators) -- conv
        # write, apply read and backward read 2 B each
        self.z16 = bool(eng.z16 and allow_z16 and self.dz16 is not None and self.fwd.family in (ops.DS_FAM_BF16D, ops.DS_FAM_FP8D)
                        and not self.skip_apply and not self.pool_inside)
        if self.z16:
            self.z = torch.empty(self.M, self.cout, device=eng.device, dtype=torch.bfloat16)
            self.fwd.d.z_dtype = ops.DS_DTYPE_BF16
            # ... CENTRED about the statistics pivot (the previous step's mean), so that z - mean does not cancel in 8 mantissa
            # bits; the passes over such a z take mean - pivot and the matching shift (ds_bn_finalize_centered)
            self.mean_c = torch.zeros(self.cout, device=eng.device)
            self.shift_c = torch.zeros(self.cout, device=eng.device)
        auto = self.cin <= 64 and self.cout <= 64
        if (eng.bnb_on_load == 2 or (eng.bnb_on_load == 1 and auto)) and not self.trainable and self.k == 1 \
                and self.dy_parts is not None:
            self.bnb = self.dgrad.enable_bn_backward_on_load(self.mean, self.rstd, self.shift, self.coef, self.dy_parts)

    def _refresh_weights(self):
        """The prepared filter forms (G g G^T for the Winograd kernels, the bf16 / fp8 / three-piece K-loop orders):
        redone when the weights changed -- every step for a trainable layer (Adam moves them), once per load for a
        frozen one."""
        eng = self.eng
        if not self.trainable and self.u_version == eng.weights_version:
            return
        for plan in (self.fwd, self.dgrad):
            if plan is not None:
                plan.prepare(self.w_ptr)
        self.u_version = eng.weights_version

    # x_ptr: input activations [B,H,W,ldx]; segs: where relu(bn(conv)) is scattered (None: the consumer, a max
    # pool, applies BatchNorm + ReLU to its own output instead -- PoolStage.forward)
    def forward(self, x_ptr, ldx, segs, x_dtype=ops.DS_DTYPE_F32, x_amax=None, defer_finalize=False):
        """defer_finalize: the caller runs ds_bn_finalize for this layer (MixedStage: one ds_bn_finalize_multi launch for the
        three convs that close the block)."""
        eng = self.eng
        plan = self.fwd
        if not self.fold:
            plan.d.ldx = ldx
        if x_dtype != ops.DS_DTYPE_F32 and not plan.x16_ok:
            raise RuntimeError("16-bit activation storage needs ds_conv_bf16 / ds_conv_fp8 for %s" % self.key)
        plan.d.x_dtype = x_dtype if plan.x16_ok else ops.DS_DTYPE_F32
        self._refresh_weights()
        if self.stem_direct:
            x_ptr = ops._p(eng.images)
        amax_p = None
        if plan.family == ops.DS_FAM_FP8D:      # per-tensor scale of the input from max|x| in a device word
            if x_amax is None:                  # no producer tracked it: one pass over x
                x_amax = self.amax[0]
                ops.absmax(x_ptr, self.B * self.H * self.W * ldx, x_amax, x_dtype)
            amax_p = ops._p(x_amax)
        if eng.training:       # batch statistics (slim.batch_norm is_training=True)
            # the column sums are taken about a pivot near the mean -- the previous step's batch mean, the
            # moving mean before the first step (bind) -- so channels with |mean| >> std keep their variance
            plan.d.flags = DS_EPI_STATS
            fin = None
            if self.fin is not None and not eng.sync_bn:      # ds_bn_finalize runs inside the conv launch (plan_finalize)
                f = self.fin
                f.count = self.M
                f.moving_mean = self.mm.data_ptr() if eng.update_moving else None
                f.moving_var = self.mv.data_ptr() if eng.update_moving else None
                fin = C.addressof(f)
            plan.run(x_ptr, self.w_ptr, ops._p(self.z), stats=ops._p(self.stats_buf), pivot=ops._p(self.mean), x_amax=amax_p, fin=fin)
            count = self.M
            if eng.sync_bn:       # statistics of the GLOBAL batch: every rank's partials are about the same pivot, so they add
                eng.all_reduce(self.stats_buf[:2 * self.cout * plan.partials])
                count = self.M * eng.sync_world
            if fin is None and not defer_finalize:
                mm, mv = (self.mm, self.mv) if eng.update_moving else (None, None)
                if eng.fuse_fin_apply and not eng.sync_bn and segs is not None and not self.skip_apply and not getattr(self, "z16", False):
                    # ds_bn_finalize and the apply pass that reads its result as ONE launch (no dependent-launch boundary)
                    ops.bn_finalize_apply_relu(self.stats_buf, plan.partials, count, self.cout, self.beta, BN_EPS, BN_DECAY,
                                               self.mean, self.rstd, self.shift, mm, mv, self.mean, self.z, self.M, segs,
                                               self.fa_ticket[0:2])
                    return
                if self.z16:
                    ops.bn_finalize_centered(self.stats_buf, plan.partials, count, self.cout, self.beta, BN_EPS, BN_DECAY, self.mean,
                                             self.rstd, self.shift, mm, mv, self.mean, self.mean_c, self.shift_c)
                else:
                    ops.bn_finalize(self.stats_buf, plan.partials, count, self.cout, self.beta, BN_EPS, BN_DECAY, self.mean,
                                    self.rstd, self.shift, mm, mv, pivot=self.mean)
        else:                  # moving statistics (is_training=False: evaluate_* on the validation split)
            plan.d.flags = 0
            # (z16: z centred about the moving mean; with it the shift is beta itself: infer_prepare on a zero mean)
            plan.run(x_ptr, self.w_ptr, ops._p(self.z), x_amax=amax_p, pivot=ops._p(self.mm) if self.z16 else None)
            ops.bn_infer_prepare(self.beta, self.mm, self.mv, BN_EPS, self.cout, self.rstd, self.shift)
            if self.z16:
                ops.bn_infer_prepare(self.beta, eng.zeros, self.mv, BN_EPS, self.cout, self.rstd, self.shift_c)
        if segs is not None and not self.skip_apply:
            ops.bn_apply_relu(self.z, self.M, self.cout, self.rstd, self.zshift, segs)

    def backward_pooled(self, pool, x_ptr=None, ldx=0, dx_ptr=None, need_dx=True):
        """Backward of conv -> BN -> ReLU -> 3x3/2 max pool from the pool's OUTPUT gradient: the pool's
        full-resolution input gradient is never written (ds_bn_pool_bwd_reduce / _apply rebuild it per patch)."""
        eng = self.eng
        M, Cc = self.M, self.cout
        if self.gbeta is None and not need_dx and not self.trainable:
            return
        B, H, W = self.B, self.OH, self.OW
        if self.part_pool[0] is not None:        # both sums from the pooled tensors: z is not read (see _bn_bwd_sums)
            self._bn_bwd_sums()
        elif self.pool_inside:
            raise RuntimeError("%s: the pooled stem needs BatchNorm's backward sums from the pooled tensors (bwd_sums)" % self.key)
        else:
            ops.bn_pool_bwd_reduce(self.z, pool.dout, pool.argmax, B, H, W, Cc, self.mean, self.rstd, self.shift,
                                   self.bwdp_buf)
            self._finalize_plain(self.pool_P)
        if not (need_dx or self.trainable):
            return
        assert not self.pool_inside          # (a frozen stem: nothing below it)
        ops.bn_pool_bwd_apply(self.z, pool.dout, pool.argmax, B, H, W, Cc, self.mean, self.rstd, self.shift, self.coef,
                              self.z)
        self._dz_amax_live = False
        if self.trainable:
            self.wgrad.d.ldx = ldx
            self.wgrad.run(x_ptr, ops._p(self.z), self.gw_ptr, ops._p(self.ws_buf), eng.ws_bytes)
        if need_dx:
            self._run_dgrad(dx_ptr)

    def _run_dgrad(self, dx_ptr, use16=False):
        sums = ops._p(self.dx_sums) if self.dx_sums is not None else None
        y = ops._p(self.dx_y) if self.dx_sums is not None else None
        am = None
        if self.dgrad.family == ops.DS_FAM_FP8D:
            am = self.dz_amax if self._dz_amax_live else self.amax[1]
            if not self._dz_amax_live:          # dz came from a kernel that does not track max|dz| (pooled BatchNorm backward)
                ops.absmax(self.z, self.M * self.cout, am)
        if use16:                # dz sits in its own bf16 tensor (backward(): ds_bn_bwd_apply_bf16)
            self.dgrad.d.x_dtype = ops.DS_DTYPE_BF16
            self.dgrad.run(ops._p(self.dz16), self.w_ptr, dx_ptr, mask=y, stats=sums, x_amax=ops._p(am))
            return
        self.dgrad.d.x_dtype = ops.DS_DTYPE_F32
        self.dgrad.run(ops._p(self.z), self.w_ptr, dx_ptr, mask=y, stats=sums, x_amax=ops._p(am))

    def backward(self, x_ptr=None, ldx=0, dx_ptr=None, need_dx=True):
        eng = self.eng
        M, Cc = self.M, self.cout
        dy_segs = self.dy_segs
        if self.gbeta is None and not need_dx and not self.trainable:
            return
        from_parts = any(ps is not None for ps in self.part_sums) or any(pp is not None for pp in self.part_pool)
        track = self.dgrad is not None and self.dgrad.family == ops.DS_FAM_FP8D
        dz = self.z if self.dz16 is None else self.dz16          # dz over z, or into its own bf16 tensor
        if eng.fuse_fin_apply and not eng.sync_bn and (need_dx or self.trainable) and not self.bnb and not self.dy2 and not self.z16:
            # the finalize and the apply pass behind it as ONE launch (ds_bn_bwd_finalize_apply)
            if from_parts:
                sg = self._sum_plan()
                self._run_reduce_jobs()
            else:
                ops.bn_bwd_reduce(self.z, dy_segs, M, Cc, self.zmean, self.rstd, self.zshift, self.bwdp_buf, ldz=self.ldz)
                if self._plain_segs is None:
                    sg = ops.SumSegments()
                    sg.nseg = 1
                    sg.c_begin[0], sg.c_end[0], sg.P[0], sg.kind[0] = 0, Cc, self.bwd_P, 0
                    sg.s[0], sg.q[0] = self.bwdp_buf.data_ptr(), self.bwdp_buf.data_ptr() + 4 * Cc * self.bwd_P
                    self._plain_segs = sg
                sg = self._plain_segs
            ops.bn_bwd_finalize_apply(sg, M, Cc, self.beta, self.gbeta, self.coef, self.z, dy_segs, self.mean, self.rstd,
                                      self.shift, dz, self.fa_ticket[2:4], amax=self.dz_amax if track else None, ldz=self.ldz)
        else:
            if from_parts:
                self._bn_bwd_sums()
            else:
                ops.bn_bwd_reduce(self.z, dy_segs, M, Cc, self.zmean, self.rstd, self.zshift, self.bwdp_buf, ldz=self.ldz)
                self._finalize_plain(self.bwd_P)
            if not (need_dx or self.trainable):
                return
            if self.bnb:              # z stays as it is: the dgrad's loader forms dz
                self._run_dgrad(dx_ptr)
                return
            ops.bn_bwd_apply(self.z, dy_segs, M, Cc, self.zmean, self.rstd, self.zshift, self.coef, dz,
                             amax=self.dz_amax if track else None, ldz=self.ldz)
        self._dz_amax_live = track
        if self.trainable:
            self.wgrad.d.ldx = ldx
            if eng.wgrad_stream is not None:
                # Conv2DBackpropFilter is a leaf of the backward graph (only the optimiser and the all-reduce wait for
                # it): off the chain that the next layer's dgrad waits on, onto the weight-gradient stream
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())          # dz is complete here
                eng.wgrad_stream.wait_event(ev)
                with torch.cuda.stream(eng.wgrad_stream):
                    self.wgrad.run(x_ptr, ops._p(self.z), self.gw_ptr, ops._p(eng.ws_set[3]), eng.ws_bytes)
                eng.wgrad_pending = True
            else:
                self.wgrad.run(x_ptr, ops._p(self.z), self.gw_ptr, ops._p(self.ws_buf), eng.ws_bytes)
        if need_dx:
            self._run_dgrad(dx_ptr, use16=self.dz16 is not None)


class Stage:
    """A node of the tower whose output is a dense NHWC tensor `out` with gradient buffer `dout`."""
    name = ""
    out = None
    dout = None


class InputStage(Stage):
    def __init__(self, eng, size):
        self.eng, self.H, self.W, self.C = eng, size, size, 4
        self.name = "input"

    def alloc(self, B):
        self.out = torch.zeros(B, self.H, self.W, 4, device=self.eng.device)


class ConvStage(Stage):
    def __init__(self, eng, name, prev, k, stride, cout, trainable, beta_bucket):
        self.eng, self.name, self.prev = eng, name, prev
        fold = isinstance(prev, InputStage)
        cin = 3 if fold else prev.C
        self.layer = ConvBN(eng, [("InceptionV1/" + name, 0, cout)], k, stride, cin, cout, prev.H, prev.W, trainable,
                            beta_bucket, fold=fold)
        self.H, self.W, self.C = self.layer.OH, self.layer.OW, cout
        self.layers = [self.layer]
        self.fused_into_pool = False
        self.pool = None

    def alloc(self, B):
        dev = self.eng.device
        eng = self.eng
        nxt = getattr(self, "next", None)
        # Conv2d_1a_7x7 -> MaxPool_2a_3x3 in one kernel (ds_conv_stem_pool / _bf16): a frozen stem whose BatchNorm + ReLU run
        # behind the pool anyway (fuse_bn_pool); its backward sums come from the pooled tensors (PoolStage.alloc)
        self.layer.pool_inside = bool(self.layer.fold and eng.stem_pool and eng.stem_direct and eng.fuse_bn_pool
                                      and not eng.mul3 and not self.layer.trainable
                                      and isinstance(nxt, PoolStage) and nxt.k == 3 and nxt.stride == 2)
        self.layer.alloc(B)
        self.out16 = self.eng.act16 and not self.layer.fold       # Conv2d_2b / 2c (the stem's output is read by hip tests only)
        self.out = torch.empty(B, self.H, self.W, self.C, device=dev, dtype=torch.bfloat16 if self.out16 else torch.float32)
        self.dout = torch.empty(B, self.H, self.W, self.C, device=dev)
        self.out_amax = self.eng.new_amax()
        self.segs = make_segments([(0, self.C, self.out.data_ptr(), self.C, ops.act_dtype(self.out),
                                    ops._p(self.out_amax))])
        self.layer.set_dy_parts([(0, self.C, self.dout.data_ptr(), self.C)])
        # the layer behind a pool that holds raw window maxima (the pooled stem) applies BatchNorm + ReLU as it loads
        self.norm_in = None
        if getattr(self.prev, "raw", False):
            assert self.layer.fwd.norm_supported()           # (PoolStage.alloc checked it with the same plan arguments)
            self.norm_in = self.prev.rs
            self.layer.fwd.d.norm_rstd, self.layer.fwd.d.norm_shift = self.norm_in[0].data_ptr(), self.norm_in[1].data_ptr()
        if not self.layer.fold:
            self.layer.make_dgrad(self.prev.C)
            # Conv2d_2c's dgrad writes the gradient of Conv2d_2b's activation: it can emit 2b's BatchNorm sums
            if isinstance(self.prev, ConvStage) and not self.prev.layer.fold:
                src = self.layer.emit_dx_sums(self.prev.out)
                if src is not None:
                    self.prev.layer.part_sums[0] = (src[0], src[1], 0, self.prev.C)
            # ... and Conv2d_2b's dgrad writes the gradient of MaxPool_2a's output.  Behind the pooled stem that IS the stem's
            # whole backward input (sum over windows of dpool (ypool > 0), ConvBN._bn_bwd_sums): its epilogue emits the sums
            # (y rebuilt from the raw window maxima where the pool's output holds those) and the reduce pass over the pooled
            # tensors -- the last launch on the backward chain -- disappears
            if isinstance(self.prev, PoolStage) and self.prev.zmax is not None and eng.stem_sums_from_dgrad:
                stem = self.prev.prev.layer
                src = self.layer.emit_dx_sums(self.prev.out)
                if src is not None:
                    if self.prev.raw:
                        self.layer.dgrad.d.mask_rstd, self.layer.dgrad.d.mask_shift = stem.rstd.data_ptr(), stem.shift.data_ptr()
                    stem.part_sums[0] = (src[0], src[1], 0, self.prev.C)
                    stem._sum_segs = None

    def forward(self):
        # fused_into_pool: this conv feeds nothing but the next max pool, which then reads z and applies BN + ReLU
        # after pooling (a quarter of the elements); `out` is not produced
        self.layer.forward(ops._p(self.prev.out), self.prev.C, None if self.fused_into_pool else self.segs,
                           ops.act_dtype(self.prev.out), getattr(self.prev, "out_amax", None))

    def backward(self, need_dx):
        need_dx = need_dx and not self.layer.fold
        dx = ops._p(self.prev.dout) if need_dx else None
        if self.fused_into_pool and self.pool.stride == 2:
            self.layer.backward_pooled(self.pool, ops._p(self.prev.out), self.prev.C, dx, need_dx)
        else:
            self.layer.backward(ops._p(self.prev.out), self.prev.C, dx, need_dx)


class PoolStage(Stage):
    def __init__(self, eng, name, prev, k, stride):
        self.eng, self.name, self.prev, self.k, self.stride = eng, name, prev, k, stride
        self.H, _ = same_pad(prev.H, k, stride)
        self.W, _ = same_pad(prev.W, k, stride)
        self.C = prev.C
        self.layers = []

    def alloc(self, B):
        dev = self.eng.device
        self.B = B
        # storage follows the input's (a pool copies values); behind a conv fused into it (reads z) the engine's choice
        p = self.prev
        o16 = p.out.dtype == torch.bfloat16 or (self.eng.act16 and isinstance(p, ConvStage) and self.k == 3)
        # Behind the pooled stem (ConvBN.pool_inside) this stage has no kernel: its output IS the stem's z buffer, the window
        # maxima.  raw: the consumer (Conv2d_2b: wide 1x1 kernel) applies relu(rstd * . + shift) as it loads; where that kernel
        # is not the one chosen (a handful of samples) a BatchNorm-apply pass over the pooled map produces the activation
        inside = isinstance(p, ConvStage) and p.layer.pool_inside
        self.raw, self.rs, self.zmax = False, None, None
        if inside:
            self.zmax = p.layer.z.view(B, self.H, self.W, self.C)
            nxt = getattr(self, "next", None)
            if isinstance(nxt, ConvStage) and nxt.layer.k == 1 and not nxt.layer.trainable and not o16:
                probe = ops.LayerPlan(ops.DS_CONV_FWD, self.eng.arith, self.eng.plan_options(), B, self.H, self.W, self.C,
                                      nxt.layer.cout, 1, 1, self.C, nxt.layer.cout, DS_EPI_STATS)
                self.raw = probe.norm_supported()
            self.rs = (p.layer.rstd, p.layer.shift)
        if self.raw:
            self.out = self.zmax
        else:
            self.out = torch.empty(B, self.H, self.W, self.C, device=dev, dtype=torch.bfloat16 if o16 else torch.float32)
        self.apply_segs = None           # (built at the first forward pass: the fp8 configuration prunes the max|.| records after alloc)
        self.dout = torch.empty(B, self.H, self.W, self.C, device=dev)
        self.argmax = torch.empty(B, self.H, self.W, self.C, dtype=torch.uint8, device=dev)
        self._own_amax = self.eng.new_amax()
        # the layers whose activation feeds nothing but this pool take their BatchNorm backward sums from the pooled
        # tensors (ConvBN._bn_bwd_sums)
        if self.eng.bwd_sums or inside:          # (the pooled stem has nothing else to take them from, whatever the switch says)
            if isinstance(p, ConvStage) and self.k == 3 and self.stride == 2:
                targets = [(p.layer, 0)]
            elif not self.eng.bwd_sums:
                targets = []
            elif isinstance(p, MixedStage):
                pb0, _, pb1b, _, pb2b, _ = p.b
                targets = [(p.fused, 0), (p.c1, pb0), (p.c2, pb0 + pb1b), (p.c3, pb0 + pb1b + pb2b)]
            else:
                targets = []
            for layer, off in targets:
                layer.part_pool[0] = (self, off)
                layer._sum_segs = None

    @property
    def out_amax(self):
        # behind a conv fused into the pool the pool kernel produces the activation and tracks its maximum; a plain
        # pool copies values, so its input's maximum bounds its output's
        if getattr(self.prev, "fused_into_pool", False):
            return self._own_amax
        return getattr(self.prev, "out_amax", None)

    def forward(self):
        p = self.prev
        if self.zmax is not None:        # the stem kernel pooled already
            if not self.raw:
                if self.apply_segs is None:
                    am = self._own_amax if getattr(self, "track_amax", True) else None
                    self.apply_segs = make_segments([(0, self.C, self.out.data_ptr(), self.C, ops.act_dtype(self.out), ops._p(am))])
                ops.bn_apply_relu(p.layer.z, self.B * self.H * self.W, self.C, self.rs[0], self.rs[1], self.apply_segs)
            return
        if getattr(p, "fused_into_pool", False):
            ops.maxpool_bn_relu_fwd(p.layer.z, p.layer.rstd, p.layer.shift, self.out, self.argmax, self.B, p.H, p.W, p.C,
                                    self.k, self.stride, amax=self._own_amax if getattr(self, "track_amax", True) else None)
        elif getattr(p, "zcat", False):      # the block's concat holds pre-BatchNorm values: normalise on load
            ops.maxpool_bn_relu_fwd(p.out, p.rs_cat[0], p.rs_cat[1], self.out, self.argmax, self.B, p.H, p.W, p.C,
                                    self.k, self.stride)
        else:
            if p.out.dtype != self.out.dtype:
                raise RuntimeError("%s: input and output storage differ (fuse_bn_pool switched after alloc?)" % self.name)
            ops.maxpool_fwd(p.out, self.out, self.argmax, self.B, p.H, p.W, p.C, self.k, self.stride, "SAME")

    def backward(self, need_dx):
        p = self.prev
        if getattr(p, "fused_into_pool", False) and self.stride == 2:
            return          # the conv in front consumes self.dout / self.argmax directly (ConvBN.backward_pooled)
        if need_dx:
            ops.maxpool_bwd(self.dout, self.argmax, p.dout, False, self.B, p.H, p.W, p.C, self.k, self.stride, "SAME")


class MixedStage(Stage):
    """One Inception block (e.g. inception_v1.py:83-96): four branches concatenated on channels."""

    def __init__(self, eng, name, prev, b0, b1, b2, b3, trainable, beta_bucket):
        self.eng, self.name, self.prev = eng, name, prev
        (b1a, b1b), (b2a, b2b) = b1, b2
        self.b = (b0, b1a, b1b, b2a, b2b, b3)
        self.H, self.W, self.C = prev.H, prev.W, b0 + b1b + b2b + b3
        cin = prev.C
        pre = "InceptionV1/%s/" % name
        b2b_scope = "Conv2d_0a_3x3" if name == "Mixed_5b" else "Conv2d_0b_3x3"   # reference quirk, :221
        nf = b0 + b1a + b2a
        self.fused = ConvBN(eng, [(pre + "Branch_0/Conv2d_0a_1x1", 0, b0),
                                  (pre + "Branch_1/Conv2d_0a_1x1", b0, b0 + b1a),
                                  (pre + "Branch_2/Conv2d_0a_1x1", b0 + b1a, nf)],
                            1, 1, cin, nf, self.H, self.W, trainable, beta_bucket)
        self.c1 = ConvBN(eng, [(pre + "Branch_1/Conv2d_0b_3x3", 0, b1b)], 3, 1, b1a, b1b, self.H, self.W, trainable,
                         beta_bucket)
        self.c2 = ConvBN(eng, [(pre + "Branch_2/" + b2b_scope, 0, b2b)], 3, 1, b2a, b2b, self.H, self.W, trainable,
                         beta_bucket)
        self.c3 = ConvBN(eng, [(pre + "Branch_3/Conv2d_0b_1x1", 0, b3)], 1, 1, cin, b3, self.H, self.W, trainable,
                         beta_bucket)
        self.layers = [self.fused, self.c1, self.c2, self.c3]
        self.c2.slot, self.c3.slot = 1, 2
        self.ev = None

    def alloc(self, B):
        eng, dev = self.eng, self.eng.device
        b0, b1a, b1b, b2a, b2b, b3 = self.b
        cin, Ct = self.prev.C, self.C
        self.B = B
        M = B * self.H * self.W
        for l in self.layers:
            l.alloc(B)
        # 16-bit activation storage (eng.act16): not for what an fp32 wgrad or the average pool reads -- Mixed_5b's
        # output, Mixed_5c's reduce outputs and output
        out16 = eng.act16 and self.name not in ("Mixed_5b", "Mixed_5c")
        r16 = eng.act16 and self.name != "Mixed_5c"
        a16 = lambda f: torch.bfloat16 if f else torch.float32
        self.out = torch.empty(B, self.H, self.W, Ct, device=dev, dtype=a16(out16))
        self.dout = torch.empty(B, self.H, self.W, Ct, device=dev)
        self.r1 = torch.empty(M, b1a, device=dev, dtype=a16(r16))
        self.r2 = torch.empty(M, b2a, device=dev, dtype=a16(r16))
        self.dr1 = torch.empty(M, b1a, device=dev)
        self.dr2 = torch.empty(M, b2a, device=dev)
        self.dpooled = torch.empty(M, cin, device=dev)
        self.argmax = torch.empty(M, cin, dtype=torch.uint8, device=dev)
        # Branch_3 = MaxPool_0a_3x3 -> Conv2d_0b_1x1 (inception_v1.py:94-95 ... :246-247) as ONE launch: the 1x1 conv's loader
        # takes the 3x3 maximum of the block input as it reads it (ds_conv_desc.pool_argmax) and records the winners; the
        # pooled tensor is never written.  Frozen fp32 layers only (a weight gradient reads the pooled activation: Mixed_5c
        # keeps the pool pass)
        self.fuse_b3 = bool(eng.fuse_branch3 and not self.c3.trainable and eng.dtype == "f32" and not eng.act16
                            and self.c3.fwd.enable_pool3(self.argmax))
        if self.fuse_b3:
            eng.need_stats(self.c3.fwd.partials * 2 * b3)          # (the fused launch groups its partial sums by image rows)
        self.pooled = None if self.fuse_b3 else torch.empty(M, cin, device=dev, dtype=self.prev.out.dtype)       # a pool copies values
        o, do = self.out.data_ptr(), self.dout.data_ptr()
        off1, off2, off3 = b0, b0 + b1b, b0 + b1b + b2b
        nf = b0 + b1a + b2a
        es, dt_o, dt_r = self.out.element_size(), ops.act_dtype(self.out), ops.act_dtype(self.r1)
        self.out_amax, self.r1_amax, self.r2_amax = eng.new_amax(), eng.new_amax(), eng.new_amax()
        am_o, am_1, am_2 = ops._p(self.out_amax), ops._p(self.r1_amax), ops._p(self.r2_amax)
        self.seg_f = make_segments([(0, b0, o, Ct, dt_o, am_o), (b0, b0 + b1a, self.r1.data_ptr(), b1a, dt_r, am_1),
                                    (b0 + b1a, nf, self.r2.data_ptr(), b2a, dt_r, am_2)])
        self.seg_1 = make_segments([(0, b1b, o + es * off1, Ct, dt_o, am_o)])
        self.seg_2 = make_segments([(0, b2b, o + es * off2, Ct, dt_o, am_o)])
        self.seg_3 = make_segments([(0, b3, o + es * off3, Ct, dt_o, am_o)])
        self.fused.set_dy_parts([(0, b0, do, Ct), (b0, b0 + b1a, self.dr1.data_ptr(), b1a),
                                 (b0 + b1a, nf, self.dr2.data_ptr(), b2a)])
        self.c1.set_dy_parts([(0, b1b, do + 4 * off1, Ct)])
        self.c2.set_dy_parts([(0, b2b, do + 4 * off2, Ct)])
        self.c3.set_dy_parts([(0, b3, do + 4 * off3, Ct)])
        # zcat: the Branch_1 / Branch_2 3x3 and the Branch_3 1x1 convs write z straight into their slices of the concat
        # buffer and NO BatchNorm-apply pass follows; whoever reads the concat -- the next block's fused 1x1 conv, its
        # Branch_3 pool, the stage pool, the BatchNorm-sums epilogue of the next block's fused dgrad -- applies
        # relu(z*rstd + shift) per channel as it loads (rs_cat; (1, 0) for the Branch_0 slice, which IS an activation:
        # it comes out of the fused layer's apply pass together with the two reduce outputs).  Bit-identical values.
        self.zcat = self._zcat_ok(B)
        self.rs_cat = None
        # batch_bn (zcat blocks): the three block-closing layers' z and dy are the columns [b0, Ct) of ONE pair of buffers, so
        # their BatchNorm launches go out once per block instead of once per layer and stream -- forward one ds_bn_finalize_multi
        # behind the join, backward one ds_bn_bwd_finalize_multi + one ds_bn_bwd_apply in front of the fork (per-channel
        # arithmetic unchanged: bit-identical).  7 blocks x (2 + 4) launches fewer per step.  Measured (profiles/r06_notes.md): the
        # backward half -0.05 ms at B = 128, -0.10 at B = 64, nothing at B = 256; the forward half another -0.04 at B <= 128 but
        # +0.06 ms at B = 256 -- the joint finalize waits for the LAST of the three chains, one more dependent launch on the
        # critical path per block -- so it is on up to 128 samples only (bit 0 forward, bit 1 backward)
        self.batch_bn = 0
        if self.zcat:
            self.rs_cat = torch.empty(2, Ct, device=dev)
            self.rs_cat[0].fill_(1.0)
            self.rs_cat[1].zero_()
            self.mean_cat = torch.zeros(Ct, device=dev)
            self.coef_cat = torch.empty(2, Ct - b0, device=dev)
            zc = self.out.view(M, Ct)
            for layer, off in ((self.c1, off1), (self.c2, off2), (self.c3, off3)):
                n = layer.cout
                layer.use_concat_slice(zc[:, off:off + n], Ct, self.rs_cat[0, off:off + n], self.rs_cat[1, off:off + n],
                                       self.mean_cat[off:off + n])
            self.batch_bn = (3 if B <= 128 else 2) if eng.batch_bn is None else int(eng.batch_bn)
            self._fin_jobs = {}
            self._close_plan = None
            self.fa_ticket = torch.zeros(2, dtype=torch.int32, device=dev)
        if getattr(self.prev, "zcat", False):    # this block reads a zcat concat
            self.fused.fwd.d.norm_rstd = self.prev.rs_cat[0].data_ptr()
            self.fused.fwd.d.norm_shift = self.prev.rs_cat[1].data_ptr()
            if self.fuse_b3:                     # ... and so does the pooling loader of its Branch_3 conv
                self.c3.fwd.d.norm_rstd = self.prev.rs_cat[0].data_ptr()
                self.c3.fwd.d.norm_shift = self.prev.rs_cat[1].data_ptr()
        self.fused.make_dgrad(cin, allow_z16=True)
        self.c1.make_dgrad(b1a, allow_z16=True)
        self.c2.make_dgrad(b2a, allow_z16=True)
        self.c3.make_dgrad(cin, allow_z16=True)
        # 16-bit labels: Branch_3's Conv2DBackpropInput output lives between that launch and the pool gradient only -- in bf16
        # storage (ds_conv_desc.z_dtype on the dgrad, ds_maxpool3_bwd_dy16): 2 B written and 2 B read per element instead of 4
        self.dpooled16 = bool(eng.dpooled16 and eng.act16 and self.c3.dgrad.family == ops.DS_FAM_BF16D)
        if self.dpooled16:
            self.dpooled = torch.empty(M, cin, device=dev, dtype=torch.bfloat16)
            self.c3.dgrad.d.z_dtype = ops.DS_DTYPE_BF16
        # BatchNorm backward sums from the epilogue of the dgrad that produces the gradient (DS_EPI_BNSUMS) instead of
        # a separate pass over z and dy:
        #  * the Branch_1 / Branch_2 3x3 dgrads write dr1 / dr2, the gradients of the fused 1x1 layer's reduce outputs;
        for part, (layer, r) in enumerate(((self.c1, self.r1), (self.c2, self.r2)), start=1):
            src = layer.emit_dx_sums(r)
            if src is not None:
                self.fused.part_sums[part] = (src[0], src[1], 0, layer.cin)
        #  * the fused 1x1 dgrad writes (last, accumulating onto the pool path: pool_first) the gradient of the block
        #    input = the previous block's concat output, i.e. one part of each of ITS four layers.
        p = self.prev
        # (the register-direct bf16 / fp8 dgrads have the accumulate epilogue too, but at their two workgroups per CU its
        # dependent read-add-store chain costs more than the pass it saves: bf16 step 12.5 -> 14.0 ms, profiles/r04_notes.md)
        # DS_POOL_FIRST_16=1 (A/B): also for the register-direct bf16 / fp8 dgrads -- 12.98 -> 13.45 ms at bf16 even with the
        # accumulate reads requested up front (14.0 before that): two waves per SIMD cannot hide a read-modify-write epilogue
        fams = (ops.DS_FAM_IGEMM, ops.DS_FAM_BF16D, ops.DS_FAM_FP8D) if _lib.tuning_env("DS_POOL_FIRST_16") == "1" else (ops.DS_FAM_IGEMM,)
        self.pool_first = bool(eng.pool_first and self.fused.dgrad.family in fams)
        if isinstance(p, MixedStage) and self.pool_first:
            src = self.fused.emit_dx_sums(p.out)
            if src is not None and getattr(p, "zcat", False):      # the epilogue rebuilds y from the concat's z
                self.fused.dgrad.d.mask_rstd = p.rs_cat[0].data_ptr()
                self.fused.dgrad.d.mask_shift = p.rs_cat[1].data_ptr()
            if src is not None:
                pb0, _, pb1b, _, pb2b, pb3 = p.b
                for layer, off in ((p.fused, 0), (p.c1, pb0), (p.c2, pb0 + pb1b), (p.c3, pb0 + pb1b + pb2b)):
                    layer.part_sums[0] = (src[0], src[1], off, cin)
                    layer._sum_segs = None
        #  * where the fused dgrad cannot accumulate (the 16-bit configurations' register-direct kernels) the order is the
        #    reverse: the dgrad writes, Branch_3's pool gradient is added LAST -- and that launch, which then holds the complete
        #    gradient of the previous block's output, emits the sums instead (ds_maxpool3_bwd_sums)
        #    split_dout: ... and then that launch need not wait for the dgrad at all -- the two addends stay TWO tensors
        #    (p.dout from the fused dgrad on the main chain, p.dout2 from the pool gradient inside the Branch_3 chain on its side
        #    stream), each producer emits the BatchNorm sums of its own addend (the sums are linear in the gradient:
        #    ds_bn_sum_segments.P2) and the one consumer, the previous block's ds_bn_bwd_apply, adds them as it reads
        #    (ds_segments.ptr2).  The pool gradient (a sixth of the 16-bit step's critical chain) leaves the main stream.
        self.pool_sums = None
        self.split_dout = False
        if isinstance(p, MixedStage) and not self.pool_first and eng.bwd_sums and eng.pool_sums and cin <= 1024 \
                and not getattr(p, "zcat", False):          # (a zcat concat holds z, not y: the dgrad epilogue's business)
            P = ops.maxpool3_bwd_sums_partials(B, p.W, cin)
            self.pool_sums = torch.empty(2 * cin * P, device=dev)
            # (the dgrad addend's sums: NOT from the register-direct kernels' DS_EPI_BNSUMS epilogue -- measured, it costs the
            # fused dgrads 51 -> 75 us each, bf16 step 9.75 -> 10.13 ms -- but from one ds_bn_bwd_reduce over (y, dout) behind the
            # dgrad: 6 B/element on the main chain where the accumulating pool gradient was 15; with mean = 0, rstd = 1,
            # shift = 0 its sums are sum g and sum g*y over y > 0, the DS_EPI_BNSUMS form)
            src, delta = None, 0
            if eng.split_dout and cin % 4 == 0:
                P1 = ops.bn_bwd_partials(M, cin)
                self.dgrad_sums = torch.empty(2 * cin * P1, device=dev)
                self._dgrad_sum_segs = make_segments([(0, cin, p.dout.data_ptr(), cin)])
                src = (self.dgrad_sums, P1)
                self.split_dout = True
                p.dout2 = torch.empty_like(p.dout)
                delta = p.dout2.data_ptr() - p.dout.data_ptr()
            pb0, _, pb1b, _, pb2b, pb3 = p.b
            for layer, off in ((p.fused, 0), (p.c1, pb0), (p.c2, pb0 + pb1b), (p.c3, pb0 + pb1b + pb2b)):
                if self.split_dout:
                    layer.part_sums[0] = (src[0], src[1], off, cin)
                    layer.part_sums2[0] = (self.pool_sums, P, off, cin)
                    layer.dy_segs.ptr2[0] = layer.dy_segs.ptr[0] + delta          # (part 0 of each layer lives in p.dout)
                    layer.dy2 = True
                else:
                    layer.part_sums[0] = (self.pool_sums, P, off, cin)
                layer._sum_segs = None

    # The three chains behind the block input -- [fused 1x1 -> Branch_1 3x3], [... -> Branch_2 3x3] and
    # [3x3/1 pool -> Branch_3 1x1] -- are independent: Branch_3 and then Branch_2 are issued on a side stream (fork /
    # join by events, each chain with its own scratch set), so one chain's single-workgroup finalize kernels and the
    # partly filled last round of its conv launches run under another chain's kernels.
    def _events(self):
        if self.ev is None:
            self.ev = [torch.cuda.Event() for _ in range(4)]
        return self.ev

    def _zcat_ok(self, B):
        """Can this block leave z in its concat buffer (see alloc)?  fp32 storage, nothing trainable in this block or in
        its consumer (a weight gradient reads activations), and a consumer whose loader can normalise: a MixedStage whose
        fused 1x1 conv runs on the wide kernel (ds_conv_igemm_norm_supported), or a max pool."""
        eng = self.eng
        nxt = getattr(self, "next", None)
        if not (eng.zcat and eng.dtype == "f32" and not eng.act16 and not eng.train_all):
            return False
        if any(l.trainable for l in self.layers):
            return False
        if isinstance(nxt, PoolStage):
            return True                      # (ds_maxpool_bn_relu_fwd: 3x3 rolling kernels, any other window generic)
        if isinstance(nxt, MixedStage) and not any(l.trainable for l in nxt.layers):
            nf = nxt.b[0] + nxt.b[1] + nxt.b[3]
            probe = ops.LayerPlan(ops.DS_CONV_FWD, eng.arith, eng.plan_options(), B, self.H, self.W, self.C, nf, 1, 1, self.C,
                                  nf, DS_EPI_STATS)
            return probe.norm_supported()
        return False

    def _pool_fwd(self):
        """Branch_3's 3x3/1 max pool of the block input (normalising on load when that is a zcat concat)."""
        p = self.prev
        if self.fuse_b3:                 # formed on load by the Branch_3 conv (alloc)
            return
        if getattr(p, "zcat", False):
            ops.maxpool_bn_relu_fwd(p.out, p.rs_cat[0], p.rs_cat[1], self.pooled, self.argmax, self.B, p.H, p.W, p.C, 3, 1)
        else:
            ops.maxpool_fwd(p.out, self.pooled, self.argmax, self.B, p.H, p.W, p.C, 3, 1, "SAME")

    def _batch_forward(self):
        """One ds_bn_finalize_multi for the block-closing layers?  (training with per-rank statistics, no in-launch finalize)"""
        eng = self.eng
        return bool((self.batch_bn & 1) and eng.training and not eng.sync_bn
                    and all(l.fin is None for l in (self.c1, self.c2, self.c3)))

    def _finalize_closing(self):
        eng = self.eng
        jobs = self._fin_jobs.get(eng.update_moving)
        if jobs is None:
            jobs = ops.BnFinalizeJobs([(l.stats_buf, l.fwd.partials, l.M, l.cout, l.beta, l.mean, l.mean, l.rstd, l.shift,
                                        l.mm if eng.update_moving else None, l.mv if eng.update_moving else None)
                                       for l in (self.c1, self.c2, self.c3)])
            self._fin_jobs[eng.update_moving] = jobs
        jobs.run(BN_EPS, BN_DECAY)

    def _batch_backward(self, need_dx):
        eng = self.eng
        return bool((self.batch_bn & 2) and need_dx and not eng.sync_bn
                    and all(l.dz16 is None and not l.bnb and l.dgrad.family != ops.DS_FAM_FP8D for l in (self.c1, self.c2, self.c3)))

    def _bn_backward_closing(self):
        """BatchNorm + ReLU backward of the three block-closing layers as one pass over the columns [b0, Ct) of the concat:
        their sums (from the next block's dgrad epilogue, or reduced per layer), ONE finalize, ONE apply that leaves dz over z."""
        b0 = self.b[0]
        Ct, Cb = self.C, self.C - b0
        M = self.B * self.H * self.W
        layers = (self.c1, self.c2, self.c3)
        if self._close_plan is None:
            sg = ops.SumSegments()
            sg.nseg = 3
            off = 0
            for i, l in enumerate(layers):
                ls = l._sum_plan()
                assert ls.nseg == 1
                sg.c_begin[i], sg.c_end[i] = off, off + l.cout
                sg.P[i], sg.kind[i], sg.s[i], sg.q[i] = ls.P[0], ls.kind[0], ls.s[0], ls.q[0]
                off += l.cout
            assert off == Cb
            dy = make_segments([(0, Cb, self.dout.data_ptr() + 4 * b0, Ct)])
            self._close_plan = (sg, dy, self.out.view(M, Ct)[:, b0:])
        sg, dy, z = self._close_plan
        for l in layers:
            l._run_reduce_jobs()
        if self.eng.fuse_fin_apply:      # ... and those two as one (ds_bn_bwd_finalize_apply)
            ops.bn_bwd_finalize_apply(sg, M, Cb, None, None, self.coef_cat, z, dy, self.mean_cat[b0:], self.rs_cat[0, b0:],
                                      self.rs_cat[1, b0:], z, self.fa_ticket, ldz=Ct, betas=[l.beta for l in layers],
                                      dbetas=[l.gbeta for l in layers])
        else:
            ops.bn_bwd_finalize_multi(sg, M, Cb, [l.beta for l in layers], [l.gbeta for l in layers], self.coef_cat)
            ops.bn_bwd_apply(z, dy, M, Cb, self.mean_cat[b0:], self.rs_cat[0, b0:], self.rs_cat[1, b0:], self.coef_cat, z, ldz=Ct)
        for l in layers:
            l._dz_amax_live = False

    def forward(self):
        p = self.prev
        b0, b1a, b1b, b2a, b2b, b3 = self.b
        x = ops._p(p.out)
        eng = self.eng
        dx_, dr_ = ops.act_dtype(p.out), ops.act_dtype(self.r1)
        ax_, a1_, a2_ = getattr(p, "out_amax", None), self.r1_amax, self.r2_amax      # fp8: max|.| words of the inputs
        defer = self._batch_forward()
        if not (eng.branch_streams and eng.side):
            self.fused.forward(x, p.C, self.seg_f, dx_, ax_)
            self.c1.forward(ops._p(self.r1), b1a, self.seg_1, dr_, a1_, defer)
            self.c2.forward(ops._p(self.r2), b2a, self.seg_2, dr_, a2_, defer)
            self._pool_fwd()
            self.c3.forward(x if self.fuse_b3 else ops._p(self.pooled), p.C, self.seg_3, dx_, ax_, defer)
            if defer:
                self._finalize_closing()
            return
        main = torch.cuda.current_stream()
        s1, s2 = eng.side
        e_in, e_f, e_2, e_3 = self._events()
        e_in.record(main)
        if eng.one_side_stream:
            s1 = s2
        with torch.cuda.stream(s2):
            s2.wait_event(e_in)
            self._pool_fwd()
            self.c3.forward(x if self.fuse_b3 else ops._p(self.pooled), p.C, self.seg_3, dx_, ax_, defer)
            if not eng.one_side_stream:
                e_3.record(s2)
        self.fused.forward(x, p.C, self.seg_f, dx_, ax_)
        if eng.one_side_stream == 2:        # only the Branch_3 chain on the side stream
            with torch.cuda.stream(s2):
                e_2.record(s2)
            self.c2.forward(ops._p(self.r2), b2a, self.seg_2, dr_, a2_, defer)
            self.c1.forward(ops._p(self.r1), b1a, self.seg_1, dr_, a1_, defer)
            main.wait_event(e_2)
            if defer:
                self._finalize_closing()
            return
        e_f.record(main)
        with torch.cuda.stream(s1):
            s1.wait_event(e_f)
            self.c2.forward(ops._p(self.r2), b2a, self.seg_2, dr_, a2_, defer)
            e_2.record(s1)
        self.c1.forward(ops._p(self.r1), b1a, self.seg_1, dr_, a1_, defer)
        main.wait_event(e_2)
        if not eng.one_side_stream:
            main.wait_event(e_3)
        if defer:
            self._finalize_closing()

    def backward(self, need_dx):
        p = self.prev
        b0, b1a, b1b, b2a, b2b, b3 = self.b
        x = ops._p(p.out)
        eng = self.eng
        # AddN of the two paths into the block input.  pool_first: the pool path WRITES p.dout inside the Branch_3 chain
        # (9 B/element, under the other chains' convs) and the fused 1x1 dgrad accumulates onto it in its epilogue;
        # otherwise (dgrad kernels without an accumulate epilogue) the dgrad writes and the pool path adds afterwards.
        pool_first = need_dx and self.pool_first
        keep = self.fused.dgrad.d.flags & ops.DS_EPI_BNSUMS
        self.fused.dgrad.d.flags = (ops.DS_EPI_ACCUM if pool_first else 0) | keep

        batched = self._batch_backward(need_dx)
        if batched:                  # BatchNorm backward of the three block-closing layers: two launches in front of the fork
            self._bn_backward_closing()

        def closing(layer, x_ptr, ldx, dx_ptr, ndx):
            if batched:
                layer._run_dgrad(dx_ptr)
            else:
                layer.backward(x_ptr, ldx, dx_ptr, ndx)

        def branch3():
            closing(self.c3, None if self.fuse_b3 else ops._p(self.pooled), p.C, ops._p(self.dpooled), need_dx)      # (x: weight gradient only)
            if pool_first:
                ops.maxpool_bwd(self.dpooled, self.argmax, p.dout, False, self.B, p.H, p.W, p.C, 3, 1, "SAME")
            elif need_dx and self.split_dout:       # the pool gradient into its OWN tensor, with its own sums: inside this chain
                ops.maxpool3_bwd_sums(self.dpooled, self.argmax, p.dout2, False, p.out, self.B, p.H, p.W, p.C, self.pool_sums)

        if not (eng.branch_streams and eng.side):
            branch3()
            closing(self.c1, ops._p(self.r1), b1a, ops._p(self.dr1), True)
            closing(self.c2, ops._p(self.r2), b2a, ops._p(self.dr2), True)
        else:
            main = torch.cuda.current_stream()
            s1, s2 = eng.side
            e_in, e_f, e_2, e_3 = self._events()
            e_in.record(main)
            if eng.one_side_stream:
                s1 = s2
            with torch.cuda.stream(s2):
                s2.wait_event(e_in)
                branch3()
                if not eng.one_side_stream:
                    e_3.record(s2)
            if eng.one_side_stream == 2:
                with torch.cuda.stream(s2):
                    e_2.record(s2)
                closing(self.c2, ops._p(self.r2), b2a, ops._p(self.dr2), True)
            else:
                with torch.cuda.stream(s1):
                    if not eng.one_side_stream:
                        s1.wait_event(e_in)
                    closing(self.c2, ops._p(self.r2), b2a, ops._p(self.dr2), True)
                    e_2.record(s1)
            closing(self.c1, ops._p(self.r1), b1a, ops._p(self.dr1), True)
            main.wait_event(e_2)
            if not eng.one_side_stream:
                main.wait_event(e_3)
        self.fused.backward(x, p.C, ops._p(p.dout) if need_dx else None, need_dx)
        if need_dx and self.split_dout:             # the sums of the dgrad's addend (the pool gradient's came with it: branch3)
            ops.bn_bwd_reduce(p.out, self._dgrad_sum_segs, self.B * p.H * p.W, p.C, eng.zeros, eng.ones, eng.zeros, self.dgrad_sums,
                              ldz=p.C)
        if need_dx and not pool_first and not self.split_dout:
            if self.pool_sums is not None:          # ... and the previous block's BatchNorm-backward sums (alloc)
                ops.maxpool3_bwd_sums(self.dpooled, self.argmax, p.dout, True, p.out, self.B, p.H, p.W, p.C, self.pool_sums)
            else:
                ops.maxpool_bwd(self.dpooled, self.argmax, p.dout, True, self.B, p.H, p.W, p.C, 3, 1, "SAME")


class InceptionV1Engine:
    """inception_v1(images, final_endpoint='Mixed_5c', num_classes, is_training=True, dropout_keep_prob)
    (image_model/inception_v1.py:254-309) as explicit forward()/backward() over HIP kernels."""

    def __init__(self, store, num_classes, image_size=224, dropout_keep_prob=0.8, trainable_bn_beta=True,
                 device="cuda", train_all=False, dtype="f32"):
        self.store, self.num_classes, self.keep = store, num_classes, dropout_keep_prob
        # arithmetic type of the 57 convs' forward and dgrad multiplies: "f32" = exact fp32 MFMA (the parity path),
        # "bf16" = bf16 MFMA with fp32 accumulation; storage, BatchNorm, wgrad, Logits and master weights stay fp32
        # "fp8" (BASELINE configs[4]): the 1x1 / 3x3 convs' forward (e4m3 x e4m3) and dgrad (e5m2 x e4m3) multiplies on
        # the fp8 matrix pipe with per-tensor power-of-two scales (ds_conv_fp8); what that kernel does not take (the
        # stem, odd channel counts) falls back to the bf16 kernels
        assert dtype in ("f32", "bf16", "fp8")
        self.dtype = dtype
        self.conv_dtype = ops.DS_DTYPE_F32 if dtype == "f32" else ops.DS_DTYPE_BF16
        self.trainable_bn_beta = trainable_bn_beta
        # train_all: full-tower fine-tuning -- every conv weight trainable, i.e. the
        # reference graph with the `trainable=False` of inception_v1.py:57-59 dropped; wgrad then runs for
        # all 57 convs (the stem through its zero-padded 4-channel input) and every variable sits in bucket 1
        self.train_all = train_all
        self.device = torch.device(device)
        self.update_moving = True
        self.training = True         # False: BatchNorm uses moving statistics, dropout is the identity
        self.reducer = None          # dp.GradientReducer, set by SentimentNet
        # sync_bn (optional; slim keeps statistics per clone): BatchNorm statistics and the two column means of
        # its backward formula over the GLOBAL batch -- one small all-reduce per layer and direction, W-rank numerics =
        # the one-process step on the whole batch.  Set by SentimentNet(sync_bn=True) under data parallelism.
        self.sync_bn = False
        self.sync_world = 1
        self.sync_group = None
        self.seed_dev = None         # device int64 added to the dropout seed (hipGraph replay draws fresh masks)
        # 1 (default): the Branch_3 chain, then the Branch_2 chain, on ONE side stream -- one cross-queue join per block (a join
        # costs ~17 us of idle GPU: 17.74 -> 17.58 ms/step against two side streams); 0: a side stream each; 2: Branch_3 only.
        # side_mode None = by batch size (alloc).  Round 5: 0 up to 32 samples (three chains in flight filled more CUs than a join
        # cost: B = 32 4.27 -> 4.15 ms).  Round 6, re-swept with split-K and the batched BatchNorm launches in: 2 (only the
        # Branch_3 chain beside the main one) up to 128 samples -- B = 16 3.53 (0) / 3.41 (1) / 3.31 (2), B = 32 3.86 / 3.99 / 3.71,
        # B = 64 a tie, B = 128 7.86 (1) / 7.82 (2) -- and 1 above (B = 256: 13.22 against 13.39 / 13.36; profiles/r06_notes.md).  The
        # 16-bit labels: 1 at every batch size (bf16 B = 32 3.83 / 3.68 / 3.81, B = 64 4.28 / 4.11 / 4.47, B = 128 5.80 (1) / 5.96 (2))
        self.side_mode = None
        self.one_side_stream = 1
        self.pool_first = True       # Mixed backward: Branch_3's pool gradient written first, fused dgrad accumulates (False: the reverse)
        # 16-bit activation storage (bf16 / fp8 configurations only): the post-ReLU activations that only feed convs and
        # pools are written as bf16 by ds_bn_apply_relu / the pools and read as bf16 by ds_conv_bf16 / ds_conv_fp8 --
        # half the bytes of every forward HBM-bound pass and of the convs' A fetch.  z, gradients, statistics, master
        # weights stay fp32; so do Mixed_5b's output and Mixed_5c's intermediates (inputs of the fp32 wgrads) and
        # Mixed_5c's output (average pool).  Not with train_all (every conv's wgrad reads its input in fp32).
        self.act16 = dtype in ("bf16", "fp8") and not train_all
        self.wgrad_side = True       # Mixed_5c's weight gradients on their own stream, off the dgrad chain
        self.side_w = None
        self.wgrad_stream = None
        self.bwd_sums = True         # BatchNorm backward sums from the producing dgrad's epilogue (DS_EPI_BNSUMS) where it can
        self.stem_direct = True      # Conv2d_1a_7x7 from the packed RGB batch (ds_conv_stem; False: generic kernel on a 4-channel copy)
        self.stem_pool = _lib.tuning_env("DS_STEM_POOL", "1") != "0"      # ... with MaxPool_2a inside that kernel (ds_conv_stem_pool; ConvStage.alloc)
        self.branch_streams = True   # Mixed blocks: Branch_2 and Branch_3 on side streams next to Branch_0/1 (False: one stream)
        self.side = None
        self.fp8_everywhere = _lib.tuning_env("DS_FP8_EVERYWHERE", "0") == "1"      # A/B: ds_conv_fp8 also where the bf16 kernels are faster
        self.fp8_wide_rule = _lib.tuning_env("DS_FP8_RULE", "0") == "1"             # A/B: the wider round-4 rule (a plan option)
        self.bf16_direct = True      # dtype bf16: ds_conv_bf16 where it wins (False: the LDS-staged bf16 kernel everywhere)
        # bf16 / fp8: the 3x3 input gradients through ds_conv_wino4_bf16x2 where ds_conv_plan's table prefers it (DS_WINO16=0: A/B)
        self.wino16 = _lib.tuning_env("DS_WINO16", "1") != "0"
        self.winograd = True         # 3x3 layers through ds_conv_wino where it wins (False: implicit GEMM everywhere)
        self.mul3 = _lib.tuning_env("DS_MUL3", "0") == "1"     # opt-in: forward 1x1 convs with fp32 products on the bf16 matrix cores
        self.bnb_on_load = int(_lib.tuning_env("DS_BNB", "1"))      # BatchNorm backward formed by the 1x1 dgrad's loader: see ConvBN.make_dgrad
        self.zcat = _lib.tuning_env("DS_ZCAT", "1") != "0"     # 3x3 / Branch_3 convs write z into the concat, consumers normalise on load
        # ds_bn_finalize inside the conv launch where the library can (ConvBN.plan_finalize; ds_conv_desc.fin).  Built, bit-identical
        # (tests), measured, OFF: the last arriver of a column tile re-reads up to 256 columns x 196 partials alone while the
        # separate launch spreads them over one workgroup per channel -- B = 32: 3.89 -> 4.36 ms, B = 64: 5.29 -> 5.59
        # (profiles/r06_notes.md).  DS_FUSE_FIN=1 switches it on (A/B)
        self.fuse_finalize = _lib.tuning_env("DS_FUSE_FIN", "0") == "1"
        # zcat blocks: the BatchNorm launches of the three block-closing layers once per block (MixedStage.alloc); DS_BATCH_BN=0: A/B
        # a BatchNorm finalize and the apply pass that reads its result as ONE launch, both directions (ds_bn_finalize_apply_relu,
        # ds_bn_bwd_finalize_apply: device-side ticket instead of a dependent-launch boundary).  Built, bit-identical (tests),
        # measured, OFF: a cross-XCD hand-off (write-through results, ticket, poll, coherent reads) costs ~15 us where the launch
        # boundary costs ~9 -- B = 256 13.23 -> 13.61 ms, B = 32 3.81 -> 4.11 (with release / acquire fences, which write back and
        # invalidate a whole L2 per workgroup: 17.4 / 6.6; profiles/r06_notes.md).  DS_FIN_APPLY=1 switches it on (A/B)
        self.fuse_fin_apply = _lib.tuning_env("DS_FIN_APPLY", "0") == "1"
        # bit 0: the forward finalizes, bit 1: the backward finalize + apply; None = by batch size (MixedStage.alloc)
        e = _lib.tuning_env("DS_BATCH_BN")
        self.batch_bn = int(e) if e else None
        self.pool_sums = _lib.tuning_env("DS_POOL_SUMS", "1") != "0"      # ... or from the Branch_3 pool gradient where that is the last addend (MixedStage.alloc)
        # ... which then stays a second tensor the consumer adds on load (MixedStage.alloc).  Built, tested, measured, OFF: the pool
        # gradient leaves the main chain, but the Branch_3 chain it joins becomes the longest of the block and the step moves
        # 6 B/element more -- bf16 9.74 -> 10.01 ms (10.13 with the sums from the register-direct dgrads' epilogue), every side
        # stream arrangement (profiles/r06_notes.md).  DS_SPLIT_DOUT=1 switches it on (A/B)
        self.split_dout = _lib.tuning_env("DS_SPLIT_DOUT", "0") == "1"
        # 16-bit configurations: z of the frozen Mixed-block layers in bf16 storage, centred about the statistics pivot
        # (ConvBN.make_dgrad; ds_conv_desc.z_dtype, ds_bn_finalize_centered): conv write, apply read and backward read move 2 B per
        # element of z instead of 4 -- bf16 step 9.58 -> 9.38 ms.  Like the 16-bit activation storage it is a rounding the fp64
        # reference emulation of the tests does not model; the labels' gates hold with it (profiles/r06_notes.md).  DS_Z16=0: A/B
        self.z16 = _lib.tuning_env("DS_Z16", "1") != "0"
        # ... and Branch_3's dgrad output (MixedStage.alloc).  Built, measured, OFF: -0.66 GB/step but no time (bf16 9.35 -> 9.37 ms:
        # the 2-byte stores of the register-direct epilogue and the 8-byte loads of the pool gradient are instruction-rate bound)
        self.dpooled16 = _lib.tuning_env("DS_DPOOLED16", "0") == "1"
        self.stem_sums_from_dgrad = _lib.tuning_env("DS_STEM_SUMS", "1") != "0"      # pooled stem: its BatchNorm sums from Conv2d_2b's dgrad epilogue
        self.dz16 = int(_lib.tuning_env("DS_DZ16", "2"))      # 16-bit configurations: bf16 dz for the frozen 1x1 (1) and 3x3 (2) layers (ConvBN.make_dgrad)
        self.fuse_branch3 = _lib.tuning_env("DS_FUSE_B3", "1") != "0"      # Branch_3's 3x3/1 max pool formed on load by its 1x1 conv (MixedStage.alloc)
        self.winograd4 = _lib.tuning_env("DS_WINO4", "1") != "0"      # ... and ds_conv_wino4 (F(4x4,3x3)) where it is faster
        # small per-GPU batches: the reduction of an F(4x4) launch that is one partial round of workgroups split over several
        # workgroups per output block (ds_conv_wino4_splitk; the library's launch-time model decides per layer).  DS_SPLITK=0: A/B
        self.splitk = _lib.tuning_env("DS_SPLITK", "1") != "0"
        # A/B: the text tower's forward kernels held back until the image tower has passed stage `text_gate` (index into
        # TOPOLOGY; None: both towers start together)
        e = _lib.tuning_env("DS_TEXT_GATE")
        self.text_gate = int(e) if e else None
        self.text_gate_event = None
        self.weights_version = 0     # bumped by SentimentNet.after_load(): frozen layers redo their G g G^T
        self._stats_n = self._bwdp_n = self._ws_bytes = 0
        self.B = None
        self.input = InputStage(self, image_size)
        self.stages = []
        prev = self.input
        for item in TOPOLOGY:
            kind, name = item[0], item[1]
            tr = train_all or name in TRAINABLE_ENDPOINTS
            bucket = 1 if tr else 2
            if kind == "conv":
                st = ConvStage(self, name, prev, item[2], item[3], item[4], tr, bucket)
            elif kind == "maxpool":
                st = PoolStage(self, name, prev, item[2], item[3])
            else:
                st = MixedStage(self, name, prev, item[2], item[3], item[4], item[5], tr, bucket)
            self.stages.append(st)
            prev = st
        self.last = prev
        for a, b in zip(self.stages[:-1], self.stages[1:]):
            a.next = b
        # BatchNorm + ReLU of a conv whose only consumer is a 3x3 max pool (Conv2d_1a_7x7 -> MaxPool_2a,
        # Conv2d_2c_3x3 -> MaxPool_3a) moves behind the pool; set fuse_bn_pool = False before the first forward to
        # get every end point materialised (image_model.inception_v1 does)
        self.fuse_bn_pool = True
        assert self.last.H == 7 and self.last.W == 7, "AvgPool_0a_7x7 + SpatialSqueeze need a 7x7 map (224x224 input)"
        self.feat = self.last.C
        lg = "InceptionV1/Logits/Conv2d_0c_1x1"
        store.declare(lg + "/weights", (1, 1, self.feat, num_classes), True, l2=True, bucket=1)
        store.declare(lg + "/biases", (num_classes,), True, bucket=1)
        self.lg = lg
        # first stage (from the top) below which nothing is trainable -> backward can stop there
        self.layers = [l for s in self.stages for l in s.layers]

    @property
    def arith(self):
        """DS_ARITH_* of the 57 convs' forward / dgrad multiplies (ds_conv_plan)."""
        if self.dtype == "f32":
            return ops.DS_ARITH_F32X3 if self.mul3 else ops.DS_ARITH_F32
        return ops.DS_ARITH_BF16 if self.dtype == "bf16" else ops.DS_ARITH_FP8

    def plan_options(self):
        """The A/B switches of this engine as ds_conv_plan option bits."""
        o = 0
        if not self.winograd:
            o |= ops.DS_PLAN_NO_WINO
        if not self.winograd4:
            o |= ops.DS_PLAN_NO_WINO4
        if not self.stem_direct:
            o |= ops.DS_PLAN_NO_STEM_DIRECT
        if not self.bf16_direct:
            o |= ops.DS_PLAN_NO_BF16_DIRECT
        if self.act16:
            o |= ops.DS_PLAN_ACT16
        if self.fp8_everywhere:
            o |= ops.DS_PLAN_FP8_EVERYWHERE
        if self.fp8_wide_rule:
            o |= ops.DS_PLAN_FP8_WIDE_RULE
        if not self.wino16:
            o |= ops.DS_PLAN_NO_WINO4H
        if not self.splitk:
            o |= ops.DS_PLAN_NO_SPLITK
        return o

    def all_reduce(self, t):
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.sync_group)

    def _prune_amax(self):
        """fp8: a producer raises a tensor's max|.| record (an atomic per wave in the BatchNorm-apply / pool kernels) only
        when a conv that READS the tensor runs on the fp8 kernels -- with fp8 restricted to the layers where it wins
        (ds_conv_plan) that is a handful of reduce outputs, not every activation of the tower."""
        def fp8(layer):
            return layer.fwd.family == ops.DS_FAM_FP8D

        def consumers(st):
            nxt = getattr(st, "next", None)
            if isinstance(nxt, ConvStage):
                return [nxt.layer]
            if isinstance(nxt, MixedStage):
                return [nxt.fused, nxt.c3]
            if isinstance(nxt, PoolStage):
                return consumers(nxt)
            return []

        for st in self.stages:
            need_out = any(fp8(l) for l in consumers(st))
            if isinstance(st, ConvStage) and not need_out:
                st.segs.amax[0] = None
            elif isinstance(st, PoolStage):
                st.track_amax = need_out
            elif isinstance(st, MixedStage):
                if not need_out:
                    for sg in (st.seg_f, st.seg_1, st.seg_2, st.seg_3):
                        sg.amax[0] = None
                if not fp8(st.c1):
                    st.seg_f.amax[1] = None
                if not fp8(st.c2):
                    st.seg_f.amax[2] = None

    def new_amax(self):
        """One word of the amax pool (None outside the fp8 configuration)."""
        if self.amax_pool is None:
            return None
        i = self._amax_next
        if i + ops.AMAX_FLOATS > self.amax_pool.numel():      # a short slice would let the fp8 kernels write out of bounds
            raise RuntimeError("amax pool exhausted (%d records): raise AMAX_RECORDS" % (self.amax_pool.numel() // ops.AMAX_FLOATS))
        self._amax_next += ops.AMAX_FLOATS
        return self.amax_pool[i:i + ops.AMAX_FLOATS]

    # scratch sizing (called by layers during alloc)
    def need_stats(self, n):
        self._stats_n = max(self._stats_n, n)

    def need_bwd_partials(self, n):
        self._bwdp_n = max(self._bwdp_n, n)

    def need_ws(self, nbytes):
        self._ws_bytes = max(self._ws_bytes, nbytes)

    def alloc(self, B):
        if self.B == B:
            return
        dev = self.device
        self.B = B
        self.alloc_gen = getattr(self, "alloc_gen", 0) + 1      # SentimentNet: a captured step is stale after this
        self.one_side_stream = (2 if (B <= 128 and self.dtype == "f32") else 1) if self.side_mode is None else int(self.side_mode)
        # fp8: device words that collect max|.| of the tensors the fp8 convs read (atomic max in the producing kernels,
        # zeroed at the start of every forward pass): the per-tensor scales without separate ds_absmax passes
        self.amax_pool = torch.zeros(AMAX_RECORDS * ops.AMAX_FLOATS, device=dev) if self.dtype == "fp8" else None
        self._amax_next = 0
        self.input.alloc(B)
        for s in self.stages:
            s.alloc(B)
        if self.amax_pool is not None:
            self._prune_amax()
        nc, F = self.num_classes, self.feat
        self.pooled = torch.empty(B, F, device=dev)
        self.dpooled = torch.empty(B, F, device=dev)
        self.mask = torch.ones(B, F, device=dev)
        self.logits = torch.empty(B, nc, device=dev)
        self.fc = ops.head_gemm_plan(B, F, nc, F, nc, nc, flags=DS_EPI_BIAS, device=dev)
        self.fc_dgrad = ops.head_gemm_plan(B, nc, F, nc, F, nc, transposed_w=True, device=dev)
        self.fc_wgrad = WgradPlan(B, 1, 1, F, F, 1, 1, 1, nc, nc, pad_t=0, pad_l=0, OH=1, OW=1)
        self.need_ws(self.fc_wgrad.ws_bytes)
        self.colsum_scratch = torch.empty(64 * max(nc, 4), device=dev)
        # three scratch sets: the branches of a Mixed block run on three streams (MixedStage.forward)
        self.stats_set = [torch.empty(max(self._stats_n, 4), device=dev) for _ in range(3)]
        self.bwdp_set = [torch.empty(max(self._bwdp_n, 4), device=dev) for _ in range(3)]
        self.ws_set = [torch.empty(max(self._ws_bytes // 4, 4), device=dev) for _ in range(4)]      # [3]: weight-gradient stream
        cws = max([pl.ws_bytes for l in self.layers for pl in (l.fwd, l.dgrad) if pl is not None] + [16])
        self.cws_set = [torch.empty(cws // 4, device=dev) for _ in range(3)]
        self.stats, self.bwd_partials, self.ws = self.stats_set[0], self.bwdp_set[0], self.ws_set[0]
        if self.side is None and self.device.type == "cuda":
            from . import streams
            self.side = [streams.get("side0", dev), streams.get("side1", dev)]
            self.side_w = streams.get("side0", dev)       # free while one_side_stream == 1
        self.ws_bytes = self._ws_bytes
        self.dummy = torch.empty(1024, device=dev)
        self.ones = torch.ones(1024, device=dev)
        self.zeros = torch.zeros(1024, device=dev)
        for l in self.layers:
            l.bind()
        st = self.store
        self.w_fc = _vp(st.ptr(self.lg + "/weights"))
        self.b_fc = _vp(st.ptr(self.lg + "/biases"))
        self.gw_fc = _vp(st.grad_ptr(self.lg + "/weights"))
        self.gb_fc = st.grad_view(self.lg + "/biases")

    # ------------------------------------------------------------------------------------------
    def forward(self, images, dropout_mask=None, seed=0):
        """images: [B,224,224,3] fp32 NHWC in [-1,1] (preprocess_for_eval range).  Returns the
        internal logits buffer [B,num_classes]."""
        B = images.shape[0]
        self.alloc(B)
        for a, b in zip(self.stages[:-1], self.stages[1:]):
            if isinstance(a, ConvStage):
                a.fused_into_pool = self.fuse_bn_pool and isinstance(b, PoolStage) and b.k == 3
                a.pool = b if a.fused_into_pool else None
        if images.dtype != torch.float32 or images.dim() != 4 or images.shape[-1] != 3:
            raise ValueError("images must be float32 NHWC [B, H, W, 3], got %s %s" % (images.dtype, tuple(images.shape)))
        if not images.is_contiguous():      # the stem kernel reads the packed batch through a raw pointer
            images = images.contiguous()
        self.images = images
        if self.amax_pool is not None:
            ops.fill(self.amax_pool, self._amax_next, 0.0)
        stem = self.stages[0].layer
        if not stem.stem_direct or (stem.trainable and self.training):
            # the generic stem kernel and the stem's wgrad (train_all) read a zero-padded 4-channel copy
            ops.pad_channels(images, 3, self.input.out, 4, B * self.input.H * self.input.W)
        gate = self.text_gate
        for i, s in enumerate(self.stages):
            s.forward()
            if gate is not None and i == gate:      # the text tower's stream may start here (SentimentNet.forward)
                self.text_gate_event = torch.cuda.Event()
                self.text_gate_event.record(torch.cuda.current_stream())
        last = self.last
        ops.avgpool_dropout_fwd(last.out, B, last.H * last.W, self.feat, self.keep if self.training else 1.0, seed,
                                dropout_mask, self.mask, self.pooled, seed_dev=self.seed_dev)
        self.fc.run(ops._p(self.pooled), self.w_fc, ops._p(self.logits), bias=self.b_fc)
        return self.logits

    def backward(self, dlogits):
        """dlogits [B,num_classes] -> gradients of every trainable image-tower variable in store.grad."""
        B, nc, F = self.B, self.num_classes, self.feat
        last = self.last
        dl = ops._p(dlogits)
        self.fc_wgrad.run(ops._p(self.pooled), dl, self.gw_fc, ops._p(self.ws), self.ws_bytes)
        ops.colsum(dlogits, B, nc, nc, self.colsum_scratch, self.gb_fc)
        self.fc_dgrad.run(dl, self.w_fc, ops._p(self.dpooled))
        ops.avgpool_dropout_bwd(self.dpooled, self.mask, B, last.H * last.W, F, self.keep, last.dout)
        n = len(self.stages)
        # below the last trainable stage only BatchNorm betas still need gradients
        stop = 0
        if not self.trainable_bn_beta:
            stop = min(i for i, s in enumerate(self.stages) if any(l.trainable for l in s.layers))
        self.wgrad_stream = self.side_w if (self.wgrad_side and self.side_w is not None) else None
        self.wgrad_pending = False
        for i in range(n - 1, stop - 1, -1):
            self.stages[i].backward(need_dx=(i > stop))
            if self.reducer is not None and not self.train_all and self.stages[i].name in TRAINABLE_ENDPOINTS:
                # every conv-weight gradient and the Logits gradients now sit in bucket 1 of the flat
                # gradient: its all-reduce can start while dgrad continues through the frozen blocks
                if self.wgrad_pending:      # ... once the weight-gradient stream is through: report from there
                    with torch.cuda.stream(self.wgrad_stream):
                        self.reducer.stage_done(self.stages[i].name)
                else:
                    self.reducer.stage_done(self.stages[i].name)
        if self.wgrad_pending:              # the optimiser (and the next forward pass, which rewrites z) wait for it
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
        if self.reducer is not None and self.train_all:
            self.reducer.stage_done(TRAINABLE_ENDPOINTS[0])      # whole tower trainable: bucket 1 closes with the stem
