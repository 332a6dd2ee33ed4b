"""SentimentNet: parameters + engines + one training step, shared by the three reference-shaped
front ends (image_model/im_model.py, text_model/text_embedding.py,
image_text_model/im_text_rnn_model.py in this package).

One step = what `slim.learning.train_step` runs for the reference's train_op
(im_text_rnn_model.py:124-135,152): forward, total loss (CE + L2 of every conv `weights`),
gradients of every trainable variable, BatchNorm moving-average updates, TF Adam.
Data-parallel: gradients are summed over ranks with RCCL all-reduce on two contiguous
buckets of the flat gradient buffer and scaled by 1/world inside the Adam kernel; the L2 term is
applied once, locally, after the reduction (slim/deployment/model_deploy.py:221-223,301-302).
"""
import math

import numpy as np
import torch

from . import ops
from .engine_image import InceptionV1Engine, WEIGHT_DECAY
from .engine_text import JointHeadEngine, TextHeadEngine, TextTowerEngine
from .functions import (InceptionV1Function, JointHeadFunction, SoftmaxCrossEntropyFunction, TextHeadFunction,
                        TextTowerFunction)
from .dp import GradientReducer
from .params import ParamStore

ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8      # tf.train.AdamOptimizer defaults (:134)


class SentimentNet:
    def __init__(self, mode="joint", nb_emotions=15, im_features_size=256, rnn_size=512, fc_size=512,
                 vocab_size=10000, embedding_dim=300, post_size=32, image_size=224, dropout_keep_prob=0.8,
                 trainable_bn_beta=True, device="cuda", process_group=None, overlap_comm=True,
                 concurrent_towers=True, train_all=False, trainable_embedding=False, dtype="f32",
                 force_dp_buckets=False, sync_bn=False):
        assert mode in ("joint", "image", "text")
        self.dtype = dtype
        if not torch.cuda.is_available():
            raise RuntimeError("tumblr_emotions_amd needs an MI355X (HIP) device: the training path has no CPU fallback")
        self.mode, self.nb_emotions, self.device = mode, nb_emotions, torch.device(device)
        self.store = ParamStore(device)
        self.image = self.text = self.head = None
        if mode in ("joint", "image"):
            nc = im_features_size if mode == "joint" else nb_emotions
            self.image = InceptionV1Engine(self.store, nc, image_size, dropout_keep_prob, trainable_bn_beta, device,
                                           train_all=train_all, dtype=dtype)
        if mode in ("joint", "text"):
            self.text = TextTowerEngine(self.store, vocab_size + 1, embedding_dim, rnn_size, post_size, device,
                                        trainable_embedding=trainable_embedding)
        if mode == "joint":
            self.head = JointHeadEngine(self.store, im_features_size, rnn_size, fc_size, nb_emotions, device)
        elif mode == "text":
            self.head = TextHeadEngine(self.store, rnn_size, nb_emotions, device)
        self.store.finalize()
        st = self.store
        self.leaves = {e.name: st.view(e.name).detach().requires_grad_() for e in st.entries.values() if e.trainable}
        if self.image is not None:
            self.image_param_names = [n for n in self.leaves if n.startswith("InceptionV1/")]
            self.image.param_names = self.image_param_names
            self.image_params = [self.leaves[n] for n in self.image_param_names]
        self.step = 0
        self.frozen_l2_sumsq = 0.0
        self.loss_buf = torch.zeros(1, device=self.device)
        self.l2_buf = torch.zeros(1, device=self.device)
        self.l2_scratch = torch.zeros(256, device=self.device)
        self.lr_t_dev = torch.zeros(1, device=self.device)
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.dlogits = None
        self.pg = process_group
        from . import streams
        streams.reserve(self.device)
        self.text_stream = streams.get("text", self.device) if (mode == "joint" and concurrent_towers) else None
        if self.text_stream is not None:
            # beside the image tower the persistent LSTM runs four row groups per workgroup: a quarter of the CUs for a
            # longer time instead of a 256-register wave on every SIMD that mostly waits -- the Winograd conv needs
            # whole SIMDs and could not run beside it (joint step 18.5 -> 17.9 ms; text-only keeps 1: shortest sequence).  Round 6:
            # EIGHT -- with the batch sorted by length and the masked steps skipped a workgroup's groups thin out as the steps go,
            # and one workgroup row on H / 16 CUs costs the image tower less than two (13.30 -> 13.24 ms; batches of fewer than
            # eight 32-row groups use as many as they have)
            self.text.seq_rows = 8
        self.reducer = GradientReducer(st.grad, st.n_bucket1, process_group, overlap_comm,
                                       force_buckets=force_dp_buckets)
        self.world = self.reducer.world
        # bucket 1 of the flat gradient is complete once these backward stages have run
        self.reducer.expect(*[n for n, e in (("Mixed_5c", self.image), ("text", self.text), ("head", self.head))
                              if e is not None])
        for e in (self.image, self.text, self.head):
            if e is not None:
                e.reducer = self.reducer
        # sync_bn: BatchNorm over the global batch (off by default: slim's clones -- and the DP parity tests -- keep the
        # statistics per rank, model_deploy.py:353-355)
        self.sync_bn = bool(sync_bn and self.world > 1 and self.image is not None)
        if self.sync_bn:
            if dtype != "f32":
                raise ValueError("sync_bn is implemented for the fp32 configuration")
            self.image.sync_bn, self.image.sync_world, self.image.sync_group = True, self.world, process_group
        self.logits = None
        self._graph = None           # captured training step (capture_step)
        self._graph_key = None
        self._eager_side_streams = None      # the image engine's one_side_stream before a capture narrowed it (release_graph)

    # ---- variables --------------------------------------------------------------------------------
    def initialize(self, seed=1):
        """Reference initialisers: conv trunc-normal(0.01) (inception_v1.py:26,59), Logits
        variance-scaling (inception_utils.py:67), tf.get_variable default glorot-uniform for LSTM kernel,
        W_fc, b_fc, W_softmax, b_softmax (im_text_rnn_model.py:89,98-104), zeros for beta / LSTM bias,
        moving mean 0 / variance 1; embedding N(0, 0.4) with a zero <ukn> row (:75)."""
        rng = np.random.RandomState(seed)
        sd = {}

        def trunc(shape, std):
            a = rng.standard_normal(size=shape)
            bad = np.abs(a) > 2
            while bad.any():
                a[bad] = rng.standard_normal(size=int(bad.sum()))
                bad = np.abs(a) > 2
            return (a * std).astype(np.float32)

        def glorot(shape):
            fi, fo = (shape[0], shape[0]) if len(shape) == 1 else (int(np.prod(shape[:-1])), shape[-1])
            lim = math.sqrt(6.0 / (fi + fo))
            return rng.uniform(-lim, lim, size=shape).astype(np.float32)

        for e in self.store.entries.values():
            n, shp = e.name, e.shape
            if n.endswith("/weights") and "/Logits/" in n:
                v = trunc(shp, math.sqrt(1.3 * 2.0 / (shp[0] * shp[1] * shp[2])))
            elif n.endswith("/weights"):
                v = trunc(shp, 0.01)
                if shp[2] == 4 and shp[0] == 7:
                    v[:, :, 3, :] = 0                      # stem: the zero-padded 4th input channel
            elif n.endswith("moving_variance"):
                v = np.ones(shp, np.float32)
            elif n.endswith("beta") or n.endswith("moving_mean") or n.endswith("/biases") or n.endswith("cell/bias"):
                v = np.zeros(shp, np.float32)
            elif n == "Text/W_embedding":
                v = rng.normal(0, 0.4, size=shp).astype(np.float32)
                v[-1] = 0
            else:
                v = glorot(shp)
            self.store.view(n).copy_(torch.from_numpy(v))
        self.after_load()

    def load_state_dict(self, sd, strict=True):
        """sd uses the reference's TF variable names; the stem's [7,7,3,64] weights are zero-padded to 4 inputs."""
        sd = dict(sd)
        k = "InceptionV1/Conv2d_1a_7x7/weights"
        if k in sd and np.asarray(sd[k]).shape[2] == 3:
            w = np.zeros((7, 7, 4, 64), np.float32)
            w[:, :, :3, :] = np.asarray(sd[k])
            sd[k] = w
        seen = self.store.load_state_dict(sd, strict)
        self.after_load()
        return seen

    def state_dict(self):
        sd = self.store.state_dict()
        k = "InceptionV1/Conv2d_1a_7x7/weights"
        if k in sd:
            sd[k] = sd[k][:, :, :3, :].copy()
        return sd

    def grads_state_dict(self):
        torch.cuda.synchronize()
        sd = self.store.state_dict(grads=True)
        k = "InceptionV1/Conv2d_1a_7x7/weights"
        if k in sd:                      # train_all: drop the zero-padded 4th input channel of the stem
            sd[k] = sd[k][:, :, :3, :].copy()
        return sd

    def after_load(self):
        """L2 of the frozen conv weights is a constant of the run: sum it once."""
        tot = 0.0
        for e in self.store.entries.values():
            if (not e.trainable) and e.name.endswith("/weights"):
                ops.sumsq(self.store.view(e.name), e.numel, self.l2_scratch, self.l2_buf)
                tot += float(self.l2_buf.item())
        self.frozen_l2_sumsq = tot
        if self.image is not None:
            self.image.weights_version += 1      # Winograd-transformed copies of the frozen filters are stale
        # ... and so is a captured step: the frozen layers' weight transforms are not part of the capture
        if getattr(self, "_graph", None) is not None:
            self.release_graph()

    # ---- forward / loss ---------------------------------------------------------------------------
    def forward(self, batch, dropout_mask=None, seed=0):
        """batch: dict with device tensors images [B,224,224,3] f32, texts [B,T] i64, seq_lens [B] i64."""
        L = self.leaves
        tx = im = None
        # image tower first: autograd runs later-created nodes first, so the (short) text backward runs
        # before the Inception backward and bucket 1 of the gradient is complete right after Mixed_5c
        inputs_ready = None
        if self.image is not None and self.text is not None and self.text_stream is not None:
            inputs_ready = torch.cuda.Event()
            inputs_ready.record(torch.cuda.current_stream())
        if self.image is not None:
            im = InceptionV1Function.apply(self.image, batch["images"], dropout_mask, seed, *self.image_params)
        if self.text is not None:
            # The text tower is ~130 launch-latency-bound kernels (LSTM steps): in the joint model it runs
            # on its own HIP stream, concurrently with the Inception tower; autograd replays the backward
            # of each node on its forward stream, so the BPTT overlaps the Inception backward too.
            side = self.text_stream if inputs_ready is not None else None
            if side is not None:
                main = torch.cuda.current_stream()
                side.wait_event(inputs_ready)          # NOT wait_stream: the image tower is already enqueued on main
                if self.image.text_gate is not None and self.image.text_gate_event is not None:
                    side.wait_event(self.image.text_gate_event)      # (A/B: start behind a stage of the image tower)
                with torch.cuda.stream(side):
                    tx = TextTowerFunction.apply(self.text, batch["texts"], batch["seq_lens"], L[self.text.KERNEL],
                                                 L[self.text.BIAS])
                main.wait_stream(side)
            else:
                tx = TextTowerFunction.apply(self.text, batch["texts"], batch["seq_lens"], L[self.text.KERNEL],
                                             L[self.text.BIAS])
        if self.mode == "image":
            self.logits = im
        elif self.mode == "text":
            self.logits = TextHeadFunction.apply(self.head, tx, L["W_softmax"], L["b_softmax"])
        else:
            self.logits = JointHeadFunction.apply(self.head, im, tx, L["W_fc"], L["b_fc"], L["W_softmax"],
                                                  L["b_softmax"])
        return self.logits

    def predict(self, batch, is_training=False, seed=None):
        """Forward only (evaluate_*): is_training=False -> BatchNorm moving statistics, no dropout;
        is_training=True reproduces the reference's evaluation on mode='train' (batch statistics and
        dropout stay on, im_text_rnn_model.py:65) but never touches the moving averages."""
        if self.image is not None:
            self.image.training, self.image.update_moving = is_training, False
        # evaluation on mode='train' keeps dropout on: every call draws a fresh mask (a fixed seed would apply
        # the identical mask to every evaluation batch)
        self._predict_calls = getattr(self, "_predict_calls", 0) + 1
        seed = (1 << 40) + self._predict_calls if seed is None else seed
        try:
            with torch.no_grad():
                return self.forward(batch, None, seed=self._rank_seed(seed))
        finally:
            if self.image is not None:
                self.image.training, self.image.update_moving = True, True

    def cross_entropy(self, logits, labels):
        if self.dlogits is None or self.dlogits.shape != logits.shape:
            self.dlogits = torch.empty(logits.shape, device=self.device)
        return SoftmaxCrossEntropyFunction.apply(logits, labels, self.loss_buf, self.dlogits)

    def total_loss_value(self):
        """Host value of slim.losses.get_total_loss(): CE + sum_conv wd*||W||^2/2 (syncs).  Since this read
        synchronises anyway it also checks the persistent LSTM's error words (ds_lstm_seq_status) and raises if a
        launch since the last read timed out on a workgroup hand-off -- its results, and this loss, are invalid."""
        reg = 0.0
        if self.store.n_l2 > 0 or self.frozen_l2_sumsq:
            reg = 0.5 * WEIGHT_DECAY * (self.frozen_l2_sumsq + float(self.l2_buf.item()))
        loss = float(self.loss_buf.item()) + reg
        self.check_status()
        return loss

    def check_status(self):
        """Device-side failure words of the step kernels (today: the persistent LSTM's hand-off timeouts)."""
        if self.text is not None:
            torch.cuda.synchronize(self.device)
            self.text.check_status()

    def _rank_seed(self, seed):
        return int(seed) * 4096 + self.reducer.rank

    # ---- one training step -----------------------------------------------------------------------
    def train_step(self, batch, lr, dropout_mask=None, seed=None):
        st = self.store
        self.step += 1
        t = self.step
        lr_t = lr * math.sqrt(1.0 - ADAM_B2 ** t) / (1.0 - ADAM_B1 ** t)
        # dropout stream: one seed per (step, rank) -- ranks must not share a mask pattern across their shards
        seed = self._rank_seed(self.step) if seed is None else seed
        if self._graph is not None:
            if self._graph_key == self._batch_key(batch, dropout_mask):
                # replay of the captured step: only the two per-step scalars change, and they live on the device
                self.seed_dev.fill_(seed)
                self.lr_t_dev.fill_(lr_t)
                self._graph.replay()
                return self.loss_buf.view(())
            if self._graph_key[-2:] != self._batch_key(batch, dropout_mask)[-2:]:
                self.release_graph()     # buffers re-allocated or weights reloaded since the capture: the graph is stale
        for p in self.leaves.values():
            p.grad = None
        logits = self.forward(batch, dropout_mask, seed)
        ce = self.cross_entropy(logits, batch["labels"])
        if st.n_l2 > 0:      # trainable part of the L2 loss, on the pre-update weights
            ops.sumsq(st.theta, st.n_l2, self.l2_scratch, self.l2_buf)
        self.reducer.begin_step()
        ce.backward()          # engines call reducer.stage_done(...): bucket 1 is all-reduced under the backward
        grad_scale = self.reducer.finish()
        ops.adam_tf(st.theta, st.grad, st.m, st.v, st.n_trainable_padded, st.n_l2, WEIGHT_DECAY, grad_scale, lr_t,
                    ADAM_B1, ADAM_B2, ADAM_EPS)
        return ce

    # ---- the same step as one hipGraph ---------------------------------------------------------------
    def _batch_key(self, batch, dropout_mask):
        # the engines' allocation generation is part of the key: alloc() for another batch size (a predict() in
        # between) frees every buffer the captured kernels point at, so the old graph must never be replayed
        gen = tuple(getattr(e, "alloc_gen", 0) for e in (self.image, self.text, self.head) if e is not None)
        wv = self.image.weights_version if self.image is not None else 0
        return tuple(sorted((k, v.data_ptr(), tuple(v.shape)) for k, v in batch.items())) + (
            None if dropout_mask is None else dropout_mask.data_ptr(), gen, wv)

    def _step_kernels(self, batch, dropout_mask):
        """Everything train_step enqueues between the batch and Adam -- forward, CE, L2 term, backward -- with the
        engines driven directly (the autograd nodes of functions.py are thin wrappers around exactly these calls,
        in this order, on these streams)."""
        st = self.store
        main = torch.cuda.current_stream()
        side = self.text_stream if (self.image is not None and self.text is not None) else None
        im = tx = None
        if side is not None:
            ready = torch.cuda.Event()
            ready.record(main)
        if self.image is not None:
            im = self.image.forward(batch["images"], dropout_mask, 0)
        if self.text is not None:
            if side is not None:
                side.wait_event(ready)
                with torch.cuda.stream(side):
                    tx = self.text.forward(batch["texts"], batch["seq_lens"])
                main.wait_stream(side)
            else:
                tx = self.text.forward(batch["texts"], batch["seq_lens"])
        if self.mode == "image":
            logits = im
        elif self.mode == "text":
            logits = self.head.forward(tx)
        else:
            logits = self.head.forward(im, tx)
        self.logits = logits
        if self.dlogits is None or self.dlogits.shape != logits.shape:
            raise RuntimeError("capture_step: run one eager train_step with this batch size first")
        B, C_ = logits.shape
        ops.softmax_ce(logits, batch["labels"], B, C_, 1.0, None, self.loss_buf, self.dlogits)
        if st.n_l2 > 0:
            ops.sumsq(st.theta, st.n_l2, self.l2_scratch, self.l2_buf)
        if self.mode == "image":
            self.image.backward(self.dlogits)
        elif self.mode == "text":
            self.text.backward(self.head.backward(self.dlogits))
        else:
            d_im, d_tx = self.head.backward(self.dlogits)
            if side is not None:              # BPTT next to the Inception backward, as in the eager step
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self.text.backward(d_tx)
                self.image.backward(d_im)
                main.wait_stream(side)
            else:
                self.text.backward(d_tx)
                self.image.backward(d_im)

    def capture_step(self, batch, dropout_mask=None):
        """Capture one whole training step on `batch`'s tensors (static addresses: refill them in place between
        steps) into a hipGraph; later train_step calls with the same tensors replay it -- one graph launch instead
        of ~900 kernel launches, which is what bounds the step below ~64 samples per GPU.  Per-step scalars (Adam's
        lr_t, the dropout seed) are read from device memory.  Single rank only: with data parallelism the RCCL
        all-reduce stays outside a graph and the eager step is used."""
        if self.reducer.active:
            return False
        st = self.store
        # eager warm-up (allocates every buffer, builds every plan) on a snapshot of the optimiser state, so that
        # capturing has no side effect on the variables, the Adam slots or the BatchNorm moving statistics
        keep = [b.clone() for b in (st.theta, st.m, st.v, st.frozen)]
        if self.image is not None and self.image.B != batch["images"].shape[0]:
            self.image.alloc(batch["images"].shape[0])
        if self.image is not None:
            # the small-batch default (a side stream per branch chain, side_mode 0) is for eager launches: hipStreamEndCapture
            # of this ROCm (7.2) segfaults on the joint step captured with three chains joined per block (B = 32, real dims),
            # and a replayed graph gains nothing from it (4.07 ms with one side stream) -- whatever side_mode says, a captured
            # step keeps ONE side stream; release_graph() restores the eager setting
            if self._eager_side_streams is None:
                self._eager_side_streams = self.image.one_side_stream
            if self.image.one_side_stream != 1:      # (0: three joined side streams break hipStreamEndCapture; 2: slower as a graph, 3.95 vs 3.78 ms)
                self.image.one_side_stream = 1
        # ... and of the BatchNorm pivots (each layer's previous batch mean), so that the first replayed step rounds
        # exactly like the eager step it replaces
        pivots = [] if self.image is None else [l.mean for l in self.image.layers]
        keep_pivots = [m.clone() for m in pivots]
        self.train_step(batch, 0.0, dropout_mask)
        self.step -= 1
        for b, k in zip((st.theta, st.m, st.v, st.frozen), keep):
            b.copy_(k)
        for m, k in zip(pivots, keep_pivots):
            m.copy_(k)
        if self.image is not None:
            self.image.seed_dev = self.seed_dev
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step_kernels(batch, dropout_mask)
            ops.adam_tf(st.theta, st.grad, st.m, st.v, st.n_trainable_padded, st.n_l2, WEIGHT_DECAY, 1.0, 0.0,
                        ADAM_B1, ADAM_B2, ADAM_EPS, lr_t_dev=self.lr_t_dev)
        self._graph, self._graph_key = g, self._batch_key(batch, dropout_mask)
        return True

    def release_graph(self):
        """Drop the captured step (also called by after_load(): a load changes weights the capture baked in)."""
        self._graph = self._graph_key = None
        if self.image is not None:
            self.image.seed_dev = None       # eager steps and predict() take the host seed again
            if self._eager_side_streams is not None:      # (capture_step may have narrowed it to one side stream)
                self.image.one_side_stream = self._eager_side_streams
                self._eager_side_streams = None
