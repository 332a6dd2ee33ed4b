"""PyTorch-CPU restatement (second, independent oracle) of the Deep Sentiment training step.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py); also the ``cpu_baseline`` ("port") leg of
bench.py.  PARITY: pinned for SAME-conv / BN moving stats / shapes / variable count by the
reference's known answers; **parity unpinned** for LSTM, head, CE, Adam (cross-checked against
oracle/tf_semantics.py only).

Uses stock torch CPU ops + autograd for the backward pass, so it shares no code with either
the NumPy oracle or the HIP product path.  Parameters live in a flat dict keyed by the
reference's TF variable names (weights HWIO, like the TF checkpoint).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import tf_semantics as S


def _same_pad_nchw(x, k, s, value):
    h, w = x.shape[2], x.shape[3]
    _, pt, pb = S.same_pad(h, k, s)
    _, pl, pr = S.same_pad(w, k, s)
    if pt or pb or pl or pr:
        x = F.pad(x, (pl, pr, pt, pb), value=value)
    return x


def conv2d_same(x, w_hwio, stride):
    """x NCHW, w HWIO.  TF SAME zero padding (extra bottom/right), cross-correlation."""
    k = w_hwio.shape[0]
    return F.conv2d(_same_pad_nchw(x, k, stride, 0.0), w_hwio.permute(3, 2, 0, 1), stride=stride)


def _bf16(t):
    """round to bfloat16 the way the HIP path does (fp32 value -> round-to-nearest-even bf16), kept in t's dtype"""
    return t.float().bfloat16().to(t.dtype)


class _ConvBf16Multiply(torch.autograd.Function):
    """VALID cross-correlation whose forward and input-gradient multiplies see bf16-rounded operands (fp32/fp64
    accumulation), while the weight gradient is taken on the unrounded operands -- what DS_DTYPE_BF16 of the HIP
    path computes (conv forward + dgrad on the bf16 matrix pipe, wgrad in fp32)."""

    @staticmethod
    def forward(ctx, xp, w_oihw, stride):
        ctx.save_for_backward(xp, w_oihw)
        ctx.stride = stride
        return F.conv2d(_bf16(xp), _bf16(w_oihw), stride=stride)

    @staticmethod
    def backward(ctx, dy):
        xp, w = ctx.saved_tensors
        dx = torch.nn.grad.conv2d_input(xp.shape, _bf16(w), _bf16(dy), stride=ctx.stride)
        dw = torch.nn.grad.conv2d_weight(xp, w.shape, dy, stride=ctx.stride)
        return dx, dw, None


# ---- fp8 emulation (the build's dtype='fp8' switch, BASELINE configs[4]; no reference counterpart) -----------------------
FP8_E4M3 = dict(fmax=448.0, mant=3, emin=-6)        # OCP e4m3fn (gfx950): weights, forward activations
FP8_E5M2 = dict(fmax=57344.0, mant=2, emin=-14)     # OCP e5m2: gradients


def fp8_pow2_scale(amax, fmax):
    """2^floor(log2(fmax / amax)) with the quotient taken in fp32, as ds_conv_fp8 / ds_weights_to_fp8 do."""
    amax = float(amax)
    if not amax > 0.0:
        return 1.0
    import numpy as np
    r = np.float32(fmax) / np.float32(amax)
    e = int(np.frexp(r)[1]) - 1
    return float(2.0 ** max(-60, min(60, e)))


def fp8_round(t, fmt):
    """saturate to +-fmax and round to the nearest representable value (ties to even), subnormals included; result in
    t's dtype"""
    a = t.abs().clamp(max=fmt["fmax"])
    _, e = torch.frexp(a)
    E = (e - 1).clamp(min=fmt["emin"])
    step = torch.ldexp(torch.ones_like(a), E - fmt["mant"])
    return torch.sign(t) * torch.round(a / step) * step


def fp8_quantize(t, fmt):
    """(values seen by the matrix pipe, scale): per-tensor power-of-two scale from max|t|, then fp8_round."""
    s = fp8_pow2_scale(t.abs().max(), fmt["fmax"])
    return fp8_round(t * s, fmt), s


class _ConvFp8Multiply(torch.autograd.Function):
    """VALID cross-correlation as ds_conv_fp8 computes it: forward on e4m3(x s_x) and e4m3(w s_w), input gradient on
    e5m2(dy s_dy) and e4m3(w s_w), each unscaled afterwards; the weight gradient on the unrounded operands."""

    @staticmethod
    def forward(ctx, xp, w_oihw, stride):
        ctx.save_for_backward(xp, w_oihw)
        ctx.stride = stride
        xq, sx = fp8_quantize(xp, FP8_E4M3)
        wq, sw = fp8_quantize(w_oihw, FP8_E4M3)
        return F.conv2d(xq, wq, stride=stride) / (sx * sw)

    @staticmethod
    def backward(ctx, dy):
        xp, w = ctx.saved_tensors
        wq, sw = fp8_quantize(w, FP8_E4M3)
        dq, sd = fp8_quantize(dy, FP8_E5M2)
        dx = torch.nn.grad.conv2d_input(xp.shape, wq, dq, stride=ctx.stride) / (sd * sw)
        dw = torch.nn.grad.conv2d_weight(xp, w.shape, dy, stride=ctx.stride)
        return dx, dw, None


def conv2d_same_fp8_multiply(x, w_hwio, stride):
    k = w_hwio.shape[0]
    return _ConvFp8Multiply.apply(_same_pad_nchw(x, k, stride, 0.0), w_hwio.permute(3, 2, 0, 1), stride)


class _ConvMixedMultiply(torch.autograd.Function):
    """The fp8 configuration as the build ships it since round 4: a layer's forward / input-gradient multiplies are fp8
    only where ds_conv_plan picks ds_conv_fp8 (fp8_where_it_wins below), bf16 otherwise -- chosen per direction."""

    @staticmethod
    def forward(ctx, xp, w_oihw, stride, fwd_fp8, bwd_fp8):
        ctx.save_for_backward(xp, w_oihw)
        ctx.stride, ctx.bwd_fp8 = stride, bwd_fp8
        if fwd_fp8:
            xq, sx = fp8_quantize(xp, FP8_E4M3)
            wq, sw = fp8_quantize(w_oihw, FP8_E4M3)
            return F.conv2d(xq, wq, stride=stride) / (sx * sw)
        return F.conv2d(_bf16(xp), _bf16(w_oihw), stride=stride)

    @staticmethod
    def backward(ctx, dy):
        xp, w = ctx.saved_tensors
        if ctx.bwd_fp8:
            wq, sw = fp8_quantize(w, FP8_E4M3)
            dq, sd = fp8_quantize(dy, FP8_E5M2)
            dx = torch.nn.grad.conv2d_input(xp.shape, wq, dq, stride=ctx.stride) / (sd * sw)
        else:
            dx = torch.nn.grad.conv2d_input(xp.shape, _bf16(w), _bf16(dy), stride=ctx.stride)
        dw = torch.nn.grad.conv2d_weight(xp, w.shape, dy, stride=ctx.stride)
        return dx, dw, None, None, None


def fp8_where_it_wins(k, cin, height, dgrad):
    """The launch rule of the build's fp8 configuration (tumblr_emotions_amd/csrc/conv_plan.cpp, measured per layer in
    profiles/r04_fp8_layers_b128.txt): ds_conv_fp8 for the FORWARD 3x3 convs with at least 96 input channels on maps of
    14 x 14 and larger (1.09-1.5x over the bf16 kernel); everything else -- every 1x1 conv, the narrow and the 7 x 7 3x3
    convs, every input gradient -- multiplies in bf16."""
    return (not dgrad) and k == 3 and cin >= 96 and height >= 14


def conv2d_same_bf16_multiply(x, w_hwio, stride):
    k = w_hwio.shape[0]
    return _ConvBf16Multiply.apply(_same_pad_nchw(x, k, stride, 0.0), w_hwio.permute(3, 2, 0, 1), stride)


def max_pool_same(x, k, s):
    return F.max_pool2d(_same_pad_nchw(x, k, s, float("-inf")), k, s)


def max_pool_same_with_argmax(x, k, s, argmax_nhwc):
    """SAME max pool whose winner per window is GIVEN (tap index kh*k+kw inside the padded window,
    [N,OH,OW,C]) instead of recomputed: lets a test evaluate the oracle's backward along exactly the
    arg-max / ReLU decisions another fp32 implementation took (see DeepSentimentRef.inject)."""
    xp = _same_pad_nchw(x, k, s, 0.0)
    win = xp.unfold(2, k, s).unfold(3, k, s)                       # [N,C,OH,OW,k,k]
    win = win.reshape(win.shape[0], win.shape[1], win.shape[2], win.shape[3], k * k)
    idx = torch.as_tensor(np.asarray(argmax_nhwc), dtype=torch.int64).permute(0, 3, 1, 2).unsqueeze(-1)
    return torch.gather(win, 4, idx).squeeze(-1)


def batch_norm_train(z, beta, eps=S.BN_EPS):
    mean = z.mean(dim=(0, 2, 3))
    var = ((z - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
    y = (z - mean[None, :, None, None]) * torch.rsqrt(var + eps)[None, :, None, None] + beta[None, :, None, None]
    return y, mean, var


def batch_norm_train_stored(z, pivot, beta, eps=S.BN_EPS):
    """batch_norm_train for a layer whose conv output the build keeps in 16-bit storage (ds_conv_desc.z_dtype, the 16-bit
    labels only -- no reference counterpart): the statistics come from the exact z, the value that is normalised is
    bf16(z - pivot) + pivot (pivot: the build's statistics pivot), and the gradient passes the rounding straight through, as the
    build's backward does (it uses the stored value for xhat and the ReLU mask and differentiates as if it were z)."""
    mean = z.mean(dim=(0, 2, 3))
    var = ((z - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
    piv = pivot.detach()[None, :, None, None]
    z_eff = z + ((_bf16(z.detach() - piv) + piv) - z.detach())
    y = (z_eff - mean[None, :, None, None]) * torch.rsqrt(var + eps)[None, :, None, None] + beta[None, :, None, None]
    return y, mean, var


def batch_norm_infer(z, beta, mm, mv, eps=S.BN_EPS):
    return (z - mm[None, :, None, None]) * torch.rsqrt(mv + eps)[None, :, None, None] + beta[None, :, None, None]


class DeepSentimentRef:
    """mode in {'joint','image','text'}: DeepSentiment (im_text_rnn_model.py:38-105),
    ImageModel (im_model.py:139-164), TextModel (text_embedding.py:37-86)."""

    def __init__(self, params, embedding=None, mode="joint", dtype=torch.float32,
                 trainable_bn_beta=True, is_training=True, train_all=False, trainable_embedding=False):
        """train_all / trainable_embedding are NOT reference behaviour (the reference freezes everything
        below Mixed_5c, inception_v1.py:57-59, and the embedding, im_text_rnn_model.py:82): they restate
        what plain TF autodiff would give if those `trainable=False` flags were dropped (SURVEY row 8f-4,
        full-tower fine-tuning), for the build's optional switch of the same name."""
        self.mode = mode
        self.dtype = dtype
        self.is_training = is_training
        self.p = {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in params.items()}
        self.embedding = None if embedding is None else torch.tensor(np.asarray(embedding), dtype=dtype)
        self.trainable = []
        for name in self.p:
            if self._is_trainable(name, trainable_bn_beta) or (
                    train_all and name.startswith("InceptionV1/") and name.endswith("/weights")):
                self.p[name].requires_grad_(True)
                self.trainable.append(name)
        if trainable_embedding and self.embedding is not None:
            self.embedding.requires_grad_(True)
            self.p["Text/W_embedding"] = self.embedding
            self.trainable.append("Text/W_embedding")
        self.adam_m = {n: torch.zeros_like(self.p[n]) for n in self.trainable}
        self.adam_v = {n: torch.zeros_like(self.p[n]) for n in self.trainable}
        self.step = 0
        self.bn_batch_stats = {}
        # Optional decision injection (tests only).  A deep BatchNorm/ReLU stack evaluated in fp32 takes a
        # few ReLU / arg-max decisions differently from the same stack in fp64 (pre-activations within
        # rounding of zero), and each flipped decision moves the upstream gradients by sqrt(flipped
        # fraction) ~ 1e-2 relative -- for ANY fp32 implementation, this oracle run in fp32 included
        # (scripts/oracle_fp32_spread.py).  With `inject` set the fp64 oracle follows given decisions:
        #   "relu/<scope>"   bool [N,H,W,C]: y = bn(z) * mask instead of relu(bn(z))
        #   "norelu/<scope>" True: no ReLU after this conv (it is applied after the following pool)
        #   "pool/<name>"    uint8 [N,OH,OW,C] winner tap of every window (+ optional
        #   "poolrelu/<name>" bool [N,OH,OW,C] ReLU mask applied to the pooled output)
        # so its gradients are a smooth function of the inputs and can be compared tightly.
        self.inject = None
        self.record = None          # dict: filled with this run's own decisions in the `inject` format
        # "bf16": the 57 BatchNorm convs multiply bf16-rounded operands in forward and dgrad (the build's
        # dtype='bf16' switch, BASELINE configs[4] groundwork; no reference counterpart -- the reference is fp32)
        self.conv_multiply = "f32"
        # scopes whose conv output the build stores as bf16(z - pivot) (InceptionV1Engine.z16, 16-bit labels): set by the tests
        # from the build's own plan; the first step's pivot is the moving mean (ConvBN.bind)
        self.z_storage_bf16 = None

    @staticmethod
    def _is_trainable(name, trainable_bn_beta):
        if name.endswith("moving_mean") or name.endswith("moving_variance"):
            return False
        if name.startswith("InceptionV1/"):
            if name.endswith("BatchNorm/beta"):
                return trainable_bn_beta            # beta trainable everywhere
            if "/Logits/" in name:
                return True                         # inception_v1.py:302-303 (outside the frozen scope)
            return any(("/%s/" % b) in name for b in S.TRAINABLE_BLOCKS)   # :229-231
        return True                                 # Text/rnn/*, W_fc, b_fc, W_softmax, b_softmax

    # -- towers -------------------------------------------------------------------------------
    def _cbr(self, x, scope, stride=1):
        w_ = self.p[scope + "/weights"]
        fp8_ok = w_.shape[0] in (1, 3) and stride == 1 and x.shape[1] % 8 == 0
        if self.conv_multiply == "fp8" and fp8_ok:
            # the layers ds_conv_fp8 takes (1x1 / 3x3, stride 1, Cin % 8 == 0); the rest -- the stem -- multiplies in bf16
            z = conv2d_same_fp8_multiply(x, self.p[scope + "/weights"], stride)
        elif self.conv_multiply == "fp8_auto":
            # fp8 only where the build's plan rule picks it (forward 3x3, >= 96 input channels, >= 14x14 maps)
            k, cin = w_.shape[0], w_.shape[2]
            z = _ConvMixedMultiply.apply(_same_pad_nchw(x, k, stride, 0.0), w_.permute(3, 2, 0, 1), stride,
                                         fp8_ok and fp8_where_it_wins(k, cin, x.shape[2], False),
                                         fp8_ok and fp8_where_it_wins(k, cin, x.shape[2], True))
        elif self.conv_multiply in ("bf16", "fp8"):
            z = conv2d_same_bf16_multiply(x, self.p[scope + "/weights"], stride)
        else:
            z = conv2d_same(x, self.p[scope + "/weights"], stride)
        beta = self.p[scope + "/BatchNorm/beta"]
        if self.is_training:
            if self.z_storage_bf16 and scope in self.z_storage_bf16:
                y, mean, var = batch_norm_train_stored(z, self.p[scope + "/BatchNorm/moving_mean"], beta)
            else:
                y, mean, var = batch_norm_train(z, beta)
            self.bn_batch_stats[scope] = (mean.detach(), var.detach())
        else:
            y = batch_norm_infer(z, beta, self.p[scope + "/BatchNorm/moving_mean"],
                                 self.p[scope + "/BatchNorm/moving_variance"])
        inj = self.inject
        if inj is not None:
            if inj.get("norelu/" + scope):
                return y
            m = inj.get("relu/" + scope)
            if m is not None:
                return y * torch.as_tensor(np.asarray(m)).permute(0, 3, 1, 2).to(y.dtype)
        if self.record is not None:
            self.record["relu/" + scope] = (y.detach() > 0).permute(0, 2, 3, 1).numpy()
        return torch.relu(y)

    def _pool(self, x, k, s, name):
        inj = self.inject
        if inj is not None and ("pool/" + name) in inj:
            y = max_pool_same_with_argmax(x, k, s, inj["pool/" + name])
            m = inj.get("poolrelu/" + name)
            if m is not None:
                y = y * torch.as_tensor(np.asarray(m)).permute(0, 3, 1, 2).to(y.dtype)
            return y
        y = max_pool_same(x, k, s)
        if self.record is not None:
            self.record["pool/" + name] = S.max_pool_argmax(x.detach().permute(0, 2, 3, 1).numpy(), k, s, "SAME")
            self.record["poolrelu/" + name] = (y.detach() > 0).permute(0, 2, 3, 1).numpy()
        return y

    def image_tower(self, images_nhwc, dropout_mask=None):
        net = images_nhwc.permute(0, 3, 1, 2)
        for item in S.INCEPTION_V1:
            kind, name = item[0], item[1]
            if kind == "conv":
                net = self._cbr(net, "InceptionV1/" + name, item[3])
            elif kind == "maxpool":
                net = self._pool(net, item[2], item[3], name)
            else:
                pre = "InceptionV1/%s/" % name
                nm = [n for (n, _, _, _) in S.mixed_conv_names(name)]
                b0 = self._cbr(net, pre + nm[0])
                b1 = self._cbr(self._cbr(net, pre + nm[1]), pre + nm[2])
                b2 = self._cbr(self._cbr(net, pre + nm[3]), pre + nm[4])
                b3 = self._cbr(self._pool(net, 3, 1, name + "/Branch_3"), pre + nm[5])
                net = torch.cat([b0, b1, b2, b3], dim=1)
        self.last_mixed_5c = net
        pooled = F.avg_pool2d(net, 7, 1)
        assert pooled.shape[2] == 1 and pooled.shape[3] == 1
        pooled = pooled[:, :, 0, 0]
        if dropout_mask is not None and self.is_training:
            pooled = pooled * dropout_mask / S.DROPOUT_KEEP
        w = self.p["InceptionV1/Logits/Conv2d_0c_1x1/weights"]
        return pooled @ w.reshape(w.shape[2], w.shape[3]) + self.p["InceptionV1/Logits/Conv2d_0c_1x1/biases"]

    def text_tower(self, texts, seq_lens, initial_state=None):
        x = self.embedding[texts]                                   # [B,T,D]
        kernel = self.p["Text/rnn/basic_lstm_cell/kernel"]
        bias = self.p["Text/rnn/basic_lstm_cell/bias"]
        b, t, _ = x.shape
        hsz = kernel.shape[1] // 4
        c = torch.zeros(b, hsz, dtype=self.dtype)
        h = torch.zeros(b, hsz, dtype=self.dtype)
        if initial_state is not None:      # (c, h); the reference always starts from zeros
            c, h = (torch.as_tensor(a, dtype=self.dtype) for a in initial_state)
        outs = []
        for s in range(t):
            z = torch.cat([x[:, s, :], h], dim=1) @ kernel + bias
            i, j, f, o = torch.split(z, hsz, dim=1)
            c_new = c * torch.sigmoid(f + S.FORGET_BIAS) + torch.sigmoid(i) * torch.tanh(j)
            h_new = torch.tanh(c_new) * torch.sigmoid(o)
            live = (s < seq_lens)[:, None]
            outs.append(torch.where(live, h_new, torch.zeros_like(h_new)))
            c = torch.where(live, c_new, c)
            h = torch.where(live, h_new, h)
        outs = torch.stack(outs, dim=1)
        return outs[torch.arange(b), seq_lens - 1]                  # gather_nd, :92

    def forward(self, batch, dropout_mask=None):
        g = lambda k: torch.as_tensor(batch[k])
        if self.mode == "image":
            return self.image_tower(g("images").to(self.dtype), dropout_mask)
        tx = self.text_tower(g("texts"), g("seq_lens"))
        if self.mode == "text":
            self.features = tx
            return tx @ self.p["W_softmax"] + self.p["b_softmax"]
        im = self.image_tower(g("images").to(self.dtype), dropout_mask)
        concat = torch.cat([im, tx], dim=1)
        self.features = concat
        # (kept for tests/golden/make_golden_dense_flips.py: the dense layer's ReLU decisions, im_text_rnn_model.py:98-101)
        self.dense_pre = concat @ self.p["W_fc"] + self.p["b_fc"]
        self.dense_out = torch.relu(self.dense_pre)
        return self.dense_out @ self.p["W_softmax"] + self.p["b_softmax"]

    # -- loss / step ---------------------------------------------------------------------------
    def loss(self, logits, labels):
        ce = F.cross_entropy(logits, torch.as_tensor(labels), reduction="mean")
        reg = torch.zeros((), dtype=self.dtype)
        if self.mode != "text":                                     # text-only graph has no slim conv
            for name, w in self.p.items():
                if name.startswith("InceptionV1/") and name.endswith("/weights"):
                    reg = reg + S.WEIGHT_DECAY * 0.5 * (w * w).sum()
        return ce + reg, ce

    def train_step(self, batch, lr, dropout_mask=None):
        """One slim train_step: fwd, total loss, grads of all trainables, BN moving-average
        updates, TF-Adam.  Returns dict(loss, ce, logits, grads)."""
        for n in self.trainable:
            self.p[n].grad = None
        logits = self.forward(batch, dropout_mask)
        total, ce = self.loss(logits, batch["labels"])
        total.backward()
        grads = {n: self.p[n].grad.detach().clone() for n in self.trainable if self.p[n].grad is not None}
        self.step += 1
        t = self.step
        lr_t = lr * math.sqrt(1 - S.ADAM_B2 ** t) / (1 - S.ADAM_B1 ** t)
        with torch.no_grad():
            for scope, (mean, var) in self.bn_batch_stats.items():
                mm = self.p[scope + "/BatchNorm/moving_mean"]
                mv = self.p[scope + "/BatchNorm/moving_variance"]
                mm.mul_(S.BN_DECAY).add_((1 - S.BN_DECAY) * mean)
                mv.mul_(S.BN_DECAY).add_((1 - S.BN_DECAY) * var)
            for n, g in grads.items():
                m, v = self.adam_m[n], self.adam_v[n]
                m.mul_(S.ADAM_B1).add_((1 - S.ADAM_B1) * g)
                v.mul_(S.ADAM_B2).add_((1 - S.ADAM_B2) * g * g)
                self.p[n].sub_(lr_t * m / (v.sqrt() + S.ADAM_EPS))
        return dict(loss=float(total.detach()), ce=float(ce.detach()), logits=logits.detach(), grads=grads)


    def train_step_dp(self, batches, lr, dropout_masks=None, injects=None):
        """One data-parallel step the way slim's in-graph clones define it (slim/deployment/model_deploy.py): clone c runs
        the forward pass on ITS sub-batch with its own BatchNorm batch statistics (:353-355), its loss is divided by the
        number of clones (:221-223), the regularisation term is added ONCE (:301-302), the clone gradients are summed
        (:414-444), and the BatchNorm moving averages come from the first clone's update ops (:353-355).  Then TF-Adam.
        `injects`: per-clone decision sets (see `inject`).  Returns dict(loss, ce=[per clone], logits=[per clone], grads)."""
        W = len(batches)
        for n in self.trainable:
            self.p[n].grad = None
        total = torch.zeros((), dtype=self.dtype)
        ces, logits_all, stats0 = [], [], None
        keep_inject = self.inject
        for c, batch in enumerate(batches):
            if injects is not None:
                self.inject = injects[c]
            logits = self.forward(batch, None if dropout_masks is None else dropout_masks[c])
            _, ce = self.loss(logits, batch["labels"])
            total = total + ce / W
            ces.append(float(ce.detach()))
            logits_all.append(logits.detach())
            if c == 0:
                stats0 = dict(self.bn_batch_stats)
        self.inject = keep_inject
        reg, _ = self.loss(logits_all[0], batches[0]["labels"])          # (total of clone 0) - (its CE) = the L2 term
        reg = reg - F.cross_entropy(logits_all[0], torch.as_tensor(batches[0]["labels"]), reduction="mean")
        total = total + reg
        total.backward()
        grads = {n: self.p[n].grad.detach().clone() for n in self.trainable if self.p[n].grad is not None}
        self.step += 1
        t = self.step
        lr_t = lr * math.sqrt(1 - S.ADAM_B2 ** t) / (1 - S.ADAM_B1 ** t)
        with torch.no_grad():
            for scope, (mean, var) in (stats0 or {}).items():
                mm = self.p[scope + "/BatchNorm/moving_mean"]
                mv = self.p[scope + "/BatchNorm/moving_variance"]
                mm.mul_(S.BN_DECAY).add_((1 - S.BN_DECAY) * mean)
                mv.mul_(S.BN_DECAY).add_((1 - S.BN_DECAY) * var)
            for n, g in grads.items():
                m, v = self.adam_m[n], self.adam_v[n]
                m.mul_(S.ADAM_B1).add_((1 - S.ADAM_B1) * g)
                v.mul_(S.ADAM_B2).add_((1 - S.ADAM_B2) * g * g)
                self.p[n].sub_(lr_t * m / (v.sqrt() + S.ADAM_EPS))
        return dict(loss=float(total.detach()), ce=ces, logits=logits_all, grads=grads)


def make_params(mode, rng, num_classes=15, im_features_size=256, embed_dim=300, rnn_size=512,
                fc_size=512, dtype=np.float32):
    """Random-init parameter dict with the reference's variable names/initialisers."""
    p = {}
    if mode in ("joint", "image"):
        p.update(S.init_inception_params(rng, im_features_size if mode == "joint" else num_classes, dtype))
    if mode in ("joint", "text"):
        p.update(S.init_text_params(rng, embed_dim, rnn_size, dtype))
    if mode == "joint":
        p.update(S.init_joint_head(rng, im_features_size + rnn_size, fc_size, num_classes, dtype))
    if mode == "text":
        p.update(S.init_text_head(rng, rnn_size, num_classes, dtype))
    return p
