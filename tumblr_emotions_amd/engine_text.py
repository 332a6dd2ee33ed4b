"""Text tower (frozen embedding -> BasicLSTMCell under dynamic_rnn -> last valid output) and the
dense heads, as explicit forward()/backward() over HIP kernels.

Reference: image_text_model/im_text_rnn_model.py:71-105, text_model/text_embedding.py:61-86.
MI355X-first choices:
  * the gather writes time-major rows so every LSTM step reads one contiguous [B, .] slab;
  * the input projection x_t*Wx+b is hoisted out of the recurrence into ONE GEMM over all T*B rows;
    the recurrence itself is ONE persistent launch per direction (ds_lstm_seq_fwd / _bwd: Wh partitioned
    over the workgroups as register-resident MFMA fragments, cell state in registers, h_t / dgates_t handed
    between the workgroups of a 32-row group through write-through stores and an arrival counter); hidden
    sizes the persistent kernel does not cover fall back to one GEMM + one cell launch per step;
  * the TF `kernel` [D+H,4H] is read in place: rows [0,D) are Wx, rows [D,D+H) are Wh;
  * h is carried through padded steps, so h[T] IS gather_nd(outputs, seq_len-1) (A8);
  * BPTT keeps dgates for all steps and does the two weight gradients as two big wgrad GEMMs
    over T*B rows at the end instead of T small ones;
  * the concat of image and text features is never built: the dense layer is two GEMMs
    accumulating into one output.
"""
import ctypes as C

import torch

from . import _lib, ops
from .ops import WgradPlan, gemm_plan, head_gemm_plan, DS_EPI_ACCUM, DS_EPI_BIAS, DS_EPI_MASK, DS_EPI_RELU

FORGET_BIAS = 1.0     # tf.contrib.rnn.BasicLSTMCell default (im_text_rnn_model.py:89)


def _vp(addr):
    return C.c_void_p(addr)


def _gemm_wgrad(M, K, N, lda, lddz):
    """dW[K,N] = A[M,K]^T * dZ[M,N]"""
    return WgradPlan(M, 1, 1, K, lda, 1, 1, 1, N, lddz, pad_t=0, pad_l=0, OH=1, OW=1)


class TextTowerEngine:
    KERNEL = "Text/rnn/basic_lstm_cell/kernel"
    BIAS = "Text/rnn/basic_lstm_cell/bias"
    EMB = "Text/W_embedding"

    def __init__(self, store, vocab_rows, embed_dim, rnn_size, post_size, device="cuda", trainable_embedding=False):
        self.store, self.V, self.D, self.H, self.T = store, vocab_rows, embed_dim, rnn_size, post_size
        self.device = torch.device(device)
        # reference: trainable=False (:82).  trainable_embedding=True is the optional fine-tuning switch
        # (SURVEY row 8f-4): dX = dgates * Wx^T, then a deterministic scatter-add into the table gradient.
        self.trainable_embedding = trainable_embedding
        store.declare(self.EMB, (vocab_rows, embed_dim), trainable_embedding, bucket=1)
        store.declare(self.KERNEL, (embed_dim + rnn_size, 4 * rnn_size), True, bucket=1)
        store.declare(self.BIAS, (4 * rnn_size,), True, bucket=1)
        self.B = None
        self.reducer = None          # dp.GradientReducer, set by SentimentNet
        self.persistent = True       # False: force the step-wise recurrence (A/B and tests)
        self.seq_rows = 1            # row groups per workgroup of the persistent kernels (`rows` of ds_lstm_seq_fwd/_bwd)
        # Length-sorted batches (round 6): the tower works on the batch in descending order of length (device-side sort, ids
        # permuted before the gather, h_last / its gradient permuted back) and the persistent kernels skip, per 32- / 16-row
        # group, the steps past the group's longest row (DS_LSTM_SKIP_MASKED) -- dynamic_rnn only copies state there.  Per
        # sample the results are the same bits; the weight gradients sum the rows in another order.
        self.sort_by_length = _lib.tuning_env("DS_LSTM_SORT", "1") != "0" 

    def alloc(self, B):
        if self.B == B:
            return
        dev, T, D, H = self.device, self.T, self.D, self.H
        self.B = B
        self.alloc_gen = getattr(self, "alloc_gen", 0) + 1
        self.use_seq = self.persistent and ops.lstm_seq_supported(B, H)
        self.sorted = bool(self.use_seq and self.sort_by_length)
        if self.sorted:
            self.perm = torch.empty(B, dtype=torch.int32, device=dev)
            self.len_sorted = torch.empty(B, dtype=torch.int64, device=dev)
            self.texts_sorted = torch.empty(B, T, dtype=torch.int64, device=dev)
            self.h_last = torch.empty(B, H, device=dev)          # h at the last valid step, ORIGINAL sample order
            self.dh_sorted = torch.empty(B, H, device=dev)
        self.seq_ws = torch.zeros(max(ops.lstm_seq_workspace(B, H) // 4, 4), dtype=torch.int32, device=dev)
        self.x = torch.empty(T * B, D, device=dev)                 # time-major embeddings
        self.gates = torch.empty(T, B, 4 * H, device=dev)          # pre-activations, then activations
        self.dgates = torch.empty(T, B, 4 * H, device=dev)
        self.h = torch.zeros(T + 1, B, H, device=dev)              # h[0] = zero initial state
        self.c = torch.zeros(T + 1, B, H, device=dev)
        self.dh = [torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)]
        self.dc = torch.empty(B, H, device=dev)
        st = self.store
        self.table = st.view(self.EMB)
        kptr = st.ptr(self.KERNEL)
        self.wx, self.wh = _vp(kptr), _vp(kptr + 4 * D * 4 * H)
        self.bias = _vp(st.ptr(self.BIAS))
        gk = st.grad_ptr(self.KERNEL)
        self.gwx, self.gwh = _vp(gk), _vp(gk + 4 * D * 4 * H)
        self.gbias = st.grad_view(self.BIAS)
        self.xproj = gemm_plan(T * B, D, 4 * H, D, 4 * H, 4 * H, flags=DS_EPI_BIAS)
        # The per-step GEMMs have M = B rows only: split K so that ~2 workgroups per CU exist, each
        # split writing its own slab; the cell kernels add the slabs (deterministic, no atomics).
        def nsplit(m, n, k):
            blocks = -(-m // 128) * -(-n // 32)
            return int(max(1, min(512 // max(blocks, 1), (k // 16) // 4)))
        self.sf, self.sb = nsplit(B, 4 * H, H), nsplit(B, H, 4 * H)
        self.rec_slabs = torch.empty(self.sf, B, 4 * H, device=dev)
        self.dh_slabs = torch.empty(self.sb, B, H, device=dev)
        self.rec = gemm_plan(B, H, 4 * H, H, 4 * H, 4 * H, splits=self.sf, z_split_stride=B * 4 * H)
        self.rec_dgrad = gemm_plan(B, 4 * H, H, 4 * H, H, 4 * H, transposed_w=True, splits=self.sb,
                                   z_split_stride=B * H)
        self.wgrad_x = _gemm_wgrad(T * B, D, 4 * H, D, 4 * H)
        self.wgrad_h = _gemm_wgrad(T * B, H, 4 * H, H, 4 * H)
        self.ws_bytes = max(self.wgrad_x.ws_bytes, self.wgrad_h.ws_bytes)
        self.ws = torch.empty(max(self.ws_bytes // 4, 4), device=dev)
        self.colsum_scratch = torch.empty(64 * 4 * H, device=dev)
        if self.trainable_embedding:
            self.dx = torch.empty(T * B, D, device=dev)
            self.x_dgrad = gemm_plan(T * B, 4 * H, D, 4 * H, D, 4 * H, transposed_w=True)
            self.gtable = st.grad_view(self.EMB)

    def forward(self, texts, seq_lens):
        """texts int64 [B,T] (pad id = vocab size), seq_lens int64 [B] (>= 1).  Returns h_last [B,H]
        (a view of the internal state buffer)."""
        B, T, H = texts.shape[0], self.T, self.H
        assert texts.shape[1] == T and texts.dtype == torch.int64 and seq_lens.dtype == torch.int64
        self.alloc(B)
        if self.sorted:         # everything below works on the batch in descending order of length
            if not seq_lens.is_contiguous() or not texts.is_contiguous():
                seq_lens, texts = seq_lens.contiguous(), texts.contiguous()
            ops.seq_sort_desc(seq_lens, B, T, self.perm, self.len_sorted)
            ops.permute_rows(texts, self.texts_sorted, self.perm, B, T, gather=True)
            seq_lens, texts = self.len_sorted, self.texts_sorted
        self.seq_lens, self.texts = seq_lens, texts
        ops.gather_rows(self.table, texts, self.x, B, T, self.D, time_major=True)
        self.xproj.run(ops._p(self.x), self.wx, ops._p(self.gates), bias=self.bias)
        if self.use_seq:
            rows = self.seq_rows | (_lib.DS_LSTM_SKIP_MASKED if self.sorted else 0)
            ops.lstm_seq_fwd(self.gates, self.wh, 4 * H, self.h, self.c, seq_lens, T, B, H, FORGET_BIAS, self.seq_ws, rows=rows)
            if self.sorted:
                ops.permute_rows(self.h[T], self.h_last, self.perm, B, H, gather=False)
                return self.h_last
            return self.h[T]
        slab = B * 4 * H
        for t in range(T):
            ns = 0
            if t > 0:           # h_0 = 0: the first step has no recurrent term
                self.rec.run(ops._p(self.h[t]), self.wh, ops._p(self.rec_slabs))
                ns = self.sf
            ops.lstm_cell_fwd(self.gates[t], self.c[t], self.h[t], seq_lens, t, B, H, FORGET_BIAS, self.c[t + 1],
                              self.h[t + 1], self.rec_slabs, ns, slab)
        return self.h[T]

    def check_status(self):
        """Raise if a persistent-LSTM launch since the last check timed out on a hand-off (its h / dgates are then
        invalid).  Reads two device words: call it where the caller synchronises anyway (loss read-out, logging)."""
        if self.B is not None and self.use_seq:
            ops.lstm_seq_status(self.seq_ws, self.B)

    def backward(self, dh_last):
        B, T, H = self.B, self.T, self.H
        if self.use_seq:
            rows = self.seq_rows
            if self.sorted:
                ops.permute_rows(dh_last, self.dh_sorted, self.perm, B, H, gather=True)
                dh_last, rows = self.dh_sorted, rows | _lib.DS_LSTM_SKIP_MASKED
            ops.lstm_seq_bwd(self.gates, self.wh, 4 * H, self.c, dh_last, dh_last.stride(0), self.seq_lens, T, B, H,
                             self.dgates, self.seq_ws, rows=rows)
            return self._weight_grads()
        dh, dh2 = self.dh
        ops.copy2d(dh_last, dh_last.stride(0), dh, H, B, H)
        ops.fill(self.dc, B * H, 0.0)
        ns = 0                  # d(h_{T-1}) is just the injected gradient
        for t in range(T - 1, -1, -1):
            # d(h_t) = carried part (dh) + dgates_{t+1} * Wh^T (split-K slabs from the previous iteration)
            ops.lstm_cell_bwd(self.gates[t], self.c[t + 1], self.c[t], dh, self.dc, self.seq_lens, t, B, H,
                              self.dgates[t], self.dc, dh2, self.dh_slabs, ns, B * H)
            if t > 0:
                self.rec_dgrad.run(ops._p(self.dgates[t]), self.wh, ops._p(self.dh_slabs))
                ns = self.sb
            dh, dh2 = dh2, dh
        self._weight_grads()

    def _weight_grads(self):
        B, T, H = self.B, self.T, self.H
        dg = ops._p(self.dgates)
        self.wgrad_x.run(ops._p(self.x), dg, self.gwx, ops._p(self.ws), self.ws_bytes)
        self.wgrad_h.run(ops._p(self.h), dg, self.gwh, ops._p(self.ws), self.ws_bytes)
        ops.colsum(self.dgates, T * B, 4 * H, 4 * H, self.colsum_scratch, self.gbias)
        if self.trainable_embedding:
            self.x_dgrad.run(dg, self.wx, ops._p(self.dx))
            ops.embedding_grad(self.dx, self.texts, self.gtable, B, T, self.D, time_major=True)
        if self.reducer is not None:
            self.reducer.stage_done("text")


class JointHeadEngine:
    """concat([im, tx]) -> relu(. W_fc + b_fc) -> . W_softmax + b_softmax   (im_text_rnn_model.py:95-105)"""

    def __init__(self, store, im_size, tx_size, fc_size, nb_emotions, device="cuda"):
        self.store, self.im, self.tx, self.fc, self.nc = store, im_size, tx_size, fc_size, nb_emotions
        self.device = torch.device(device)
        store.declare("W_fc", (im_size + tx_size, fc_size), True, bucket=1)
        store.declare("b_fc", (fc_size,), True, bucket=1)
        store.declare("W_softmax", (fc_size, nb_emotions), True, bucket=1)
        store.declare("b_softmax", (nb_emotions,), True, bucket=1)
        self.B = None
        self.reducer = None          # dp.GradientReducer, set by SentimentNet

    def alloc(self, B):
        if self.B == B:
            return
        dev, im, tx, fc, nc = self.device, self.im, self.tx, self.fc, self.nc
        self.B = B
        self.alloc_gen = getattr(self, "alloc_gen", 0) + 1
        self.dense = torch.empty(B, fc, device=dev)
        self.ddense = torch.empty(B, fc, device=dev)
        self.logits = torch.empty(B, nc, device=dev)
        self.d_im = torch.empty(B, im, device=dev)
        self.d_tx = torch.empty(B, tx, device=dev)
        st = self.store
        w = st.ptr("W_fc")
        self.w_im, self.w_tx = _vp(w), _vp(w + 4 * im * fc)
        gw = st.grad_ptr("W_fc")
        self.gw_im, self.gw_tx = _vp(gw), _vp(gw + 4 * im * fc)
        self.b_fc, self.gb_fc = _vp(st.ptr("b_fc")), st.grad_view("b_fc")
        self.w_sm, self.gw_sm = _vp(st.ptr("W_softmax")), _vp(st.grad_ptr("W_softmax"))
        self.b_sm, self.gb_sm = _vp(st.ptr("b_softmax")), st.grad_view("b_softmax")
        self.fc_im = None      # built on first forward (input strides are the callers')
        self.ws_bytes = 0
        self.colsum_scratch = torch.empty(64 * max(fc, nc), device=dev)

    def _plans(self, ld_im, ld_tx):
        B, im, tx, fc, nc = self.B, self.im, self.tx, self.fc, self.nc
        self.fc_im = head_gemm_plan(B, im, fc, ld_im, fc, fc, device=self.device)
        self.fc_tx = head_gemm_plan(B, tx, fc, ld_tx, fc, fc, flags=DS_EPI_ACCUM | DS_EPI_BIAS | DS_EPI_RELU, device=self.device)
        self.sm = head_gemm_plan(B, fc, nc, fc, nc, nc, flags=DS_EPI_BIAS, device=self.device)
        self.sm_dgrad = head_gemm_plan(B, nc, fc, nc, fc, nc, transposed_w=True, flags=DS_EPI_MASK, ldmask=fc, device=self.device)
        self.sm_wgrad = _gemm_wgrad(B, fc, nc, fc, nc)
        self.im_dgrad = head_gemm_plan(B, fc, im, fc, im, fc, transposed_w=True, device=self.device)
        self.tx_dgrad = head_gemm_plan(B, fc, tx, fc, tx, fc, transposed_w=True, device=self.device)
        self.im_wgrad = _gemm_wgrad(B, im, fc, ld_im, fc)
        self.tx_wgrad = _gemm_wgrad(B, tx, fc, ld_tx, fc)
        self.ws_bytes = max(p.ws_bytes for p in (self.sm_wgrad, self.im_wgrad, self.tx_wgrad))
        self.ws = torch.empty(max(self.ws_bytes // 4, 4), device=self.device)
        self._lds = (ld_im, ld_tx)

    def forward(self, im_feat, tx_feat):
        B = im_feat.shape[0]
        self.alloc(B)
        lds = (im_feat.stride(0), tx_feat.stride(0))
        if self.fc_im is None or self._lds != lds:
            self._plans(*lds)
        self.im_feat, self.tx_feat = im_feat, tx_feat
        self.fc_im.run(ops._p(im_feat), self.w_im, ops._p(self.dense))
        self.fc_tx.run(ops._p(tx_feat), self.w_tx, ops._p(self.dense), bias=self.b_fc)
        self.sm.run(ops._p(self.dense), self.w_sm, ops._p(self.logits), bias=self.b_sm)
        return self.logits

    def backward(self, dlogits):
        B, fc, nc = self.B, self.fc, self.nc
        dl, ws = ops._p(dlogits), ops._p(self.ws)
        self.sm_wgrad.run(ops._p(self.dense), dl, self.gw_sm, ws, self.ws_bytes)
        ops.colsum(dlogits, B, nc, nc, self.colsum_scratch, self.gb_sm)
        self.sm_dgrad.run(dl, self.w_sm, ops._p(self.ddense), mask=ops._p(self.dense))      # ReluGrad fused
        dd = ops._p(self.ddense)
        self.im_wgrad.run(ops._p(self.im_feat), dd, self.gw_im, ws, self.ws_bytes)
        self.tx_wgrad.run(ops._p(self.tx_feat), dd, self.gw_tx, ws, self.ws_bytes)
        ops.colsum(self.ddense, B, fc, fc, self.colsum_scratch, self.gb_fc)
        self.im_dgrad.run(dd, self.w_im, ops._p(self.d_im))
        self.tx_dgrad.run(dd, self.w_tx, ops._p(self.d_tx))
        if self.reducer is not None:
            self.reducer.stage_done("head")
        return self.d_im, self.d_tx


class TextHeadEngine:
    """logits = h_last . W_softmax + b_softmax   (text_model/text_embedding.py:84-86)"""

    def __init__(self, store, rnn_size, nb_emotions, device="cuda"):
        self.store, self.H, self.nc = store, rnn_size, nb_emotions
        self.device = torch.device(device)
        store.declare("W_softmax", (rnn_size, nb_emotions), True, bucket=1)
        store.declare("b_softmax", (nb_emotions,), True, bucket=1)
        self.B = None
        self.reducer = None          # dp.GradientReducer, set by SentimentNet

    def alloc(self, B):
        if self.B == B:
            return
        dev, H, nc = self.device, self.H, self.nc
        self.B = B
        self.alloc_gen = getattr(self, "alloc_gen", 0) + 1
        self.logits = torch.empty(B, nc, device=dev)
        self.d_tx = torch.empty(B, H, device=dev)
        st = self.store
        self.w_sm, self.gw_sm = _vp(st.ptr("W_softmax")), _vp(st.grad_ptr("W_softmax"))
        self.b_sm, self.gb_sm = _vp(st.ptr("b_softmax")), st.grad_view("b_softmax")
        self.sm = head_gemm_plan(B, H, nc, H, nc, nc, flags=DS_EPI_BIAS, device=self.device)
        self.sm_dgrad = head_gemm_plan(B, nc, H, nc, H, nc, transposed_w=True, device=self.device)
        self.sm_wgrad = _gemm_wgrad(B, H, nc, H, nc)
        self.ws_bytes = self.sm_wgrad.ws_bytes
        self.ws = torch.empty(max(self.ws_bytes // 4, 4), device=dev)
        self.colsum_scratch = torch.empty(64 * max(nc, 4), device=dev)

    def forward(self, tx_feat):
        self.alloc(tx_feat.shape[0])
        assert tx_feat.stride(0) == self.H
        self.tx_feat = tx_feat
        self.sm.run(ops._p(tx_feat), self.w_sm, ops._p(self.logits), bias=self.b_sm)
        return self.logits

    def backward(self, dlogits):
        dl = ops._p(dlogits)
        self.sm_wgrad.run(ops._p(self.tx_feat), dl, self.gw_sm, ops._p(self.ws), self.ws_bytes)
        ops.colsum(dlogits, self.B, self.nc, self.nc, self.colsum_scratch, self.gb_sm)
        self.sm_dgrad.run(dl, self.w_sm, ops._p(self.d_tx))
        if self.reducer is not None:
            self.reducer.stage_done("head")
        return self.d_tx
