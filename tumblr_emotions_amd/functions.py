"""torch.autograd.Function wrappers around the HIP engines.

The reference builds a TF graph and lets `slim.learning.create_train_op`
(image_text_model/im_text_rnn_model.py:135) derive the backward pass; here each tower is one
autograd node whose forward/backward enqueue our kernels.  Parameters are leaf tensors that alias
the flat parameter buffer; the gradients returned are views of the flat gradient buffer, so after
`loss.backward()` the whole gradient already lies contiguously for RCCL and the fused Adam.
"""
import torch

from . import ops


class InceptionV1Function(torch.autograd.Function):
    """(images, *trainable image-tower variables) -> logits [B, num_classes]."""

    @staticmethod
    def forward(ctx, engine, images, dropout_mask, seed, *params):
        ctx.engine = engine
        ctx.n = len(params)
        return engine.forward(images, dropout_mask, seed)

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.engine
        eng.backward(dlogits if dlogits.is_contiguous() else dlogits.contiguous())
        # fresh views of the flat gradient buffer: autograd's AccumulateGrad can then adopt them as
        # `.grad` without a copy (a view object that is also referenced elsewhere would be cloned)
        g = eng.store.grad_view
        return (None, None, None, None) + tuple(g(n) for n in eng.param_names)


class TextTowerFunction(torch.autograd.Function):
    """(texts, seq_lens, kernel, bias) -> h at the last valid step [B, H]."""

    @staticmethod
    def forward(ctx, engine, texts, seq_lens, kernel, bias):
        ctx.engine = engine
        return engine.forward(texts, seq_lens)

    @staticmethod
    def backward(ctx, dh):
        eng = ctx.engine
        eng.backward(dh)
        return None, None, None, eng.store.grad_view(eng.KERNEL), eng.store.grad_view(eng.BIAS)


class JointHeadFunction(torch.autograd.Function):
    """(im_feat, tx_feat, W_fc, b_fc, W_softmax, b_softmax) -> logits."""

    @staticmethod
    def forward(ctx, engine, im_feat, tx_feat, w_fc, b_fc, w_sm, b_sm):
        ctx.engine = engine
        return engine.forward(im_feat, tx_feat)

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.engine
        d_im, d_tx = eng.backward(dlogits if dlogits.is_contiguous() else dlogits.contiguous())
        g = eng.store.grad_view
        return None, d_im, d_tx, g("W_fc"), g("b_fc"), g("W_softmax"), g("b_softmax")


class TextHeadFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, tx_feat, w_sm, b_sm):
        ctx.engine = engine
        return engine.forward(tx_feat)

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.engine
        d_tx = eng.backward(dlogits if dlogits.is_contiguous() else dlogits.contiguous())
        g = eng.store.grad_view
        return None, d_tx, g("W_softmax"), g("b_softmax")


class SoftmaxCrossEntropyFunction(torch.autograd.Function):
    """slim.losses.softmax_cross_entropy(logits, one_hot(labels)) = mean_b CE_b  (:124-125)."""

    @staticmethod
    def forward(ctx, logits, labels, loss_buf, dlogits_buf):
        B, C_ = logits.shape
        ops.softmax_ce(logits, labels, B, C_, 1.0, None, loss_buf, None)
        ctx.save = (logits, labels, dlogits_buf)
        return loss_buf.view(())

    @staticmethod
    def backward(ctx, dloss):
        logits, labels, dl = ctx.save
        B, C_ = logits.shape
        ops.softmax_ce(logits, labels, B, C_, 1.0, dloss.reshape(1), None, dl)
        return dl, None, None, None
