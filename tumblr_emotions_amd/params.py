"""Flat parameter store: every trainable variable of the model lives in ONE fp32 device buffer
(`theta`), with matching Adam moments (`m`, `v`) and gradient (`grad`) buffers, so that

  * tf.train.AdamOptimizer.apply_gradients (im_text_rnn_model.py:134-135) is one fused kernel,
  * the data-parallel gradient exchange is two RCCL all-reduces over contiguous ranges
    (bucket 1 = everything that is ready when Mixed_5c's backward finishes, bucket 2 = the
    upstream BatchNorm betas), and
  * the L2 regulariser's gradient (slim/nets/inception_utils.py:63-64) is applied to a prefix.

Non-trainable variables (frozen conv weights image_model/inception_v1.py:57-59, BatchNorm moving
statistics, the embedding table im_text_rnn_model.py:82) live in a second flat buffer.

Variables keep the reference's TensorFlow names (`InceptionV1/Mixed_3b/Branch_1/Conv2d_0a_1x1/weights`
...).  The three 1x1 convs that read a block's input are stored as one horizontally fused
[1,1,Cin,b0+b1a+b2a] tensor; `state_dict` / `load_state_dict` split / join the column ranges.
"""
import numpy as np
import torch


class Entry:
    __slots__ = ("name", "shape", "numel", "offset", "trainable", "l2", "bucket", "columns")

    def __init__(self, name, shape, trainable, l2=False, bucket=1, columns=None):
        self.name, self.shape = name, tuple(shape)
        self.numel = int(np.prod(shape))
        self.trainable, self.l2, self.bucket = trainable, l2, bucket
        self.columns = columns      # [(tf_name, c0, c1)] for fused tensors, else None
        self.offset = -1


class ParamStore:
    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self.entries = {}
        self._order_tr = []
        self._order_fr = []
        self.finalized = False

    # -- declaration ------------------------------------------------------------------------------
    def declare(self, name, shape, trainable, l2=False, bucket=1, columns=None):
        assert not self.finalized and name not in self.entries, name
        e = Entry(name, shape, trainable, l2, bucket, columns)
        self.entries[name] = e
        (self._order_tr if trainable else self._order_fr).append(e)
        return e

    def finalize(self):
        """Lay out: [L2-regularised weights | other bucket-1 variables | bucket-2 variables]."""
        tr = ([e for e in self._order_tr if e.l2 and e.bucket == 1] +
              [e for e in self._order_tr if not e.l2 and e.bucket == 1] +
              [e for e in self._order_tr if e.bucket == 2])
        assert all(not (e.l2 and e.bucket == 2) for e in self._order_tr)
        off = 0
        for e in tr:
            e.offset = off
            off += (e.numel + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.n_trainable_padded = off
        self.n_l2 = sum((e.numel + 3) // 4 * 4 for e in tr if e.l2)
        self.n_bucket1 = sum((e.numel + 3) // 4 * 4 for e in tr if e.bucket == 1)
        self.n_trainable = sum(e.numel for e in tr)
        off = 0
        for e in self._order_fr:
            e.offset = off
            off += (e.numel + 3) // 4 * 4
        n_tr = max(self.n_trainable_padded, 4)
        self.theta = torch.zeros(n_tr, dtype=torch.float32, device=self.device)
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.grad = torch.zeros_like(self.theta)
        self.frozen = torch.zeros(max(off, 4), dtype=torch.float32, device=self.device)
        self.finalized = True

    # -- views ------------------------------------------------------------------------------------
    def _buf(self, e):
        return self.theta if e.trainable else self.frozen

    def view(self, name):
        e = self.entries[name]
        return self._buf(e)[e.offset:e.offset + e.numel].view(e.shape)

    def grad_view(self, name):
        e = self.entries[name]
        assert e.trainable
        return self.grad[e.offset:e.offset + e.numel].view(e.shape)

    def ptr(self, name, elem_offset=0):
        e = self.entries[name]
        return self._buf(e).data_ptr() + 4 * (e.offset + elem_offset)

    def grad_ptr(self, name, elem_offset=0):
        e = self.entries[name]
        return self.grad.data_ptr() + 4 * (e.offset + elem_offset)

    # -- (de)serialisation under the reference's TF variable names ---------------------------------
    def tf_names(self):
        out = []
        for e in self.entries.values():
            if e.columns:
                out += [n for (n, _, _) in e.columns]
            else:
                out.append(e.name)
        return out

    def slot_view(self, name, which):
        """View of an Adam slot ('m' / 'v') of a trainable variable."""
        e = self.entries[name]
        assert e.trainable
        return getattr(self, which)[e.offset:e.offset + e.numel].view(e.shape)

    def load_state_dict(self, sd, strict=True, which="value"):
        """sd: {tf_name: array-like}.  Fused tensors are assembled from their column ranges.
        which='value' loads variables; 'm' / 'v' load the Adam slots of the trainable ones."""
        seen = set()
        for e in self.entries.values():
            if which != "value" and not e.trainable:
                continue
            dst = self.view(e.name) if which == "value" else self.slot_view(e.name, which)
            if e.columns:
                for (n, c0, c1) in e.columns:
                    if n in sd:
                        src = torch.as_tensor(np.asarray(sd[n]), dtype=torch.float32)
                        dst[..., c0:c1].copy_(src.reshape(dst[..., c0:c1].shape))
                        seen.add(n)
                    elif strict:
                        raise KeyError(n)
            elif e.name in sd:
                src = torch.as_tensor(np.asarray(sd[e.name]), dtype=torch.float32)
                dst.copy_(src.reshape(dst.shape))
                seen.add(e.name)
            elif strict:
                raise KeyError(e.name)
        return seen

    def state_dict(self, grads=False):
        out = {}
        for e in self.entries.values():
            if grads and not e.trainable:
                continue
            t = (self.grad_view(e.name) if grads else self.view(e.name)).detach().cpu()
            if e.columns:
                for (n, c0, c1) in e.columns:
                    out[n] = t[..., c0:c1].contiguous().numpy()
            else:
                out[e.name] = t.numpy().copy()
        return out
