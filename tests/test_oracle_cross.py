"""Cross-check the two independent CPU restatements (NumPy with hand-written backward formulas vs
PyTorch-CPU autograd) in fp64.  These are the rules the reference pins with no test of its own
(A5, A7-A10) -- "parity unpinned" -- so agreement of two restatements plus the
known answers in test_oracle_known_answers.py is the strongest statement available here."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import tf_semantics as S
from oracle import torch_ref as R

RNG = np.random.RandomState(11)


def test_conv_backward_formulas():
    for (k, s, h) in [(3, 1, 6), (1, 1, 5), (7, 2, 12), (3, 2, 7)]:
        x = RNG.normal(size=(2, h, h, 3))
        w = RNG.normal(size=(k, k, 3, 4))
        xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_(True)
        wt = torch.tensor(w, requires_grad=True)
        y = R.conv2d_same(xt, wt, s)
        dy = RNG.normal(size=tuple(y.shape))
        y.backward(torch.tensor(dy))
        dy_nhwc = dy.transpose(0, 2, 3, 1)
        np.testing.assert_allclose(S.conv2d_same(x, w, s), y.detach().permute(0, 2, 3, 1).numpy(), atol=1e-12)
        np.testing.assert_allclose(S.conv2d_same_bwd_input(dy_nhwc, w, x.shape, s),
                                   xt.grad.permute(0, 2, 3, 1).numpy(), atol=1e-12)
        np.testing.assert_allclose(S.conv2d_same_bwd_filter(x, dy_nhwc, w.shape, s), wt.grad.numpy(), atol=1e-11)


def test_max_pool_forward_backward():
    for (k, s, h, mode) in [(3, 2, 8, "SAME"), (3, 1, 7, "SAME"), (2, 2, 6, "VALID"), (3, 2, 7, "SAME")]:
        x = RNG.normal(size=(2, h, h, 5))
        xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_(True)
        y = R.max_pool_same(xt, k, s) if mode == "SAME" else F.max_pool2d(xt, k, s)
        dy = RNG.normal(size=tuple(y.shape))
        y.backward(torch.tensor(dy))
        np.testing.assert_allclose(S.max_pool(x, k, s, mode), y.detach().permute(0, 2, 3, 1).numpy())
        np.testing.assert_allclose(S.max_pool_bwd(x, dy.transpose(0, 2, 3, 1), k, s, mode),
                                   xt.grad.permute(0, 2, 3, 1).numpy(), atol=1e-13)


def test_batch_norm_backward_formula():
    z = RNG.normal(1.0, 2.0, size=(3, 5, 5, 6))
    beta = RNG.normal(size=6)
    dy = RNG.normal(size=z.shape)
    y, mean, var, xhat, rstd = S.batch_norm_train(z, beta)
    dz, dbeta = S.batch_norm_train_bwd(dy, xhat, rstd)
    zt = torch.tensor(z).permute(0, 3, 1, 2).requires_grad_(True)
    bt = torch.tensor(beta, requires_grad=True)
    yt, _, _ = R.batch_norm_train(zt, bt)
    yt.backward(torch.tensor(dy).permute(0, 3, 1, 2))
    np.testing.assert_allclose(y, yt.detach().permute(0, 2, 3, 1).numpy(), atol=1e-12)
    np.testing.assert_allclose(dz, zt.grad.permute(0, 2, 3, 1).numpy(), atol=1e-12)
    np.testing.assert_allclose(dbeta, bt.grad.numpy(), atol=1e-12)


def test_lstm_forward_and_bptt():
    b, t, d, h = 5, 7, 6, 8
    x = RNG.normal(size=(b, t, d))
    seq = np.array([7, 1, 3, 6, 2])
    kernel = RNG.normal(0, 0.4, size=(d + h, 4 * h))
    bias = RNG.normal(0, 0.1, size=4 * h)
    outs, h_last, cache = S.lstm_forward(x, seq, kernel, bias, keep_cache=True)
    # outputs are zero past seq_len; h_last is the last valid step (dynamic_rnn semantics, A8)
    for i in range(b):
        assert np.all(outs[i, seq[i]:] == 0)
        np.testing.assert_array_equal(h_last[i], outs[i, seq[i] - 1])
    emb = np.zeros((1, d))
    ref = R.DeepSentimentRef({"Text/rnn/basic_lstm_cell/kernel": kernel, "Text/rnn/basic_lstm_cell/bias": bias,
                              "W_softmax": np.zeros((h, 2)), "b_softmax": np.zeros(2)},
                             embedding=emb, mode="text", dtype=torch.float64)
    ref.embedding = torch.tensor(x.reshape(b * t, d))
    ids = torch.arange(b * t).reshape(b, t)
    ht = ref.text_tower(ids, torch.tensor(seq))
    np.testing.assert_allclose(h_last, ht.detach().numpy(), atol=1e-13)
    dh = RNG.normal(size=(b, h))
    ht.backward(torch.tensor(dh))
    dk, db = S.lstm_backward(dh, seq, kernel, cache)
    np.testing.assert_allclose(dk, ref.p["Text/rnn/basic_lstm_cell/kernel"].grad.numpy(), atol=1e-12)
    np.testing.assert_allclose(db, ref.p["Text/rnn/basic_lstm_cell/bias"].grad.numpy(), atol=1e-12)


def test_softmax_ce_and_grad():
    z = RNG.normal(size=(9, 15)) * 3
    y = RNG.randint(0, 15, size=9)
    zt = torch.tensor(z, requires_grad=True)
    l = F.cross_entropy(zt, torch.tensor(y))
    l.backward()
    np.testing.assert_allclose(S.softmax_cross_entropy(z, y), float(l), atol=1e-13)
    np.testing.assert_allclose(S.softmax_cross_entropy_grad(z, y), zt.grad.numpy(), atol=1e-14)


def test_tf_adam_differs_from_torch_adam_and_matches_formula():
    w = RNG.normal(size=50)
    m = np.zeros(50)
    v = np.zeros(50)
    wt = w.copy()
    for t in range(1, 4):
        g = RNG.normal(size=50) * 1e-4          # small grads make the epsilon placement visible
        w, m, v = S.adam_step(w, g, m, v, t, 1e-3)
        lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        assert np.all(np.isfinite(w)) and lr_t > 0
    assert not np.allclose(w, wt)


def test_joint_step_numpy_forward_matches_torch_fp64_small():
    rng = np.random.RandomState(5)
    params = R.make_params("joint", rng, num_classes=15, im_features_size=8, embed_dim=12, rnn_size=16,
                           fc_size=10, dtype=np.float64)
    batch = S.synthetic_batch(2, 9, 40, seed=0)
    emb = S.synthetic_embedding(40, 12).astype(np.float64)
    logits_np, aux = S.deep_sentiment_forward(params, emb, batch["images"].astype(np.float64), batch["texts"],
                                              batch["seq_lens"])
    ref = R.DeepSentimentRef(params, emb, "joint", torch.float64)
    logits_t = ref.forward(batch)
    np.testing.assert_allclose(logits_np, logits_t.detach().numpy(), rtol=1e-8, atol=1e-10)
    total, ce = ref.loss(logits_t, batch["labels"])
    np.testing.assert_allclose(S.total_loss(logits_np, batch["labels"], params), float(total), rtol=1e-10)
    # one TF-Adam step on the LSTM kernel through the NumPy formula equals the torch_ref update
    k0 = params["Text/rnn/basic_lstm_cell/kernel"].copy()
    out = ref.train_step(batch, 1e-3)
    g = out["grads"]["Text/rnn/basic_lstm_cell/kernel"].numpy()
    k1, _, _ = S.adam_step(k0, g, np.zeros_like(k0), np.zeros_like(k0), 1, 1e-3)
    np.testing.assert_allclose(k1, ref.p["Text/rnn/basic_lstm_cell/kernel"].detach().numpy(), atol=1e-14)
    # trainable set: 5c weights (6) + 57 betas + Logits W,b + LSTM kernel,bias + 4 head vars
    assert len(ref.trainable) == 6 + 57 + 2 + 2 + 4
    n_tr = sum(int(np.prod(ref.p[n].shape)) for n in ref.trainable)
    assert n_tr == 1344512 + 7280 + (1024 * 8 + 8) + ((12 + 16) * 64 + 64) + (24 * 10 + 10 + 10 * 15 + 15)


def test_decision_injection_reproduces_the_plain_backward():
    """DeepSentimentRef.inject (ReLU masks / pool winners given instead of recomputed) is what lets the GPU
    parity test compare gradients tightly; fed with the oracle's OWN decisions it must reproduce the plain
    fp64 step exactly, also in the 'ReLU after the pool' form used for the two convs that only feed a pool."""
    rng = np.random.RandomState(8)
    params = R.make_params("image", rng, num_classes=15, dtype=np.float64)
    for k in params:
        if k.endswith("beta"):
            params[k] = rng.normal(0, 0.1, size=params[k].shape)
    batch = S.synthetic_batch(2, 8, 10, seed=2)
    plain = R.DeepSentimentRef(params, None, "image", torch.float64)
    plain.record = {}
    out = plain.train_step(batch, 1e-3)
    rec = plain.record
    assert sum(k.startswith("relu/") for k in rec) == 57 and sum(k.startswith("pool/") for k in rec) == 4 + 9

    inj = R.DeepSentimentRef(params, None, "image", torch.float64)
    inj.inject = {k: v for k, v in rec.items() if not k.startswith("poolrelu/")}
    out2 = inj.train_step(batch, 1e-3)
    np.testing.assert_allclose(out2["logits"].numpy(), out["logits"].numpy(), atol=1e-12)
    for n, g in out["grads"].items():
        np.testing.assert_allclose(out2["grads"][n].numpy(), g.numpy(), atol=1e-12 * max(1.0, float(g.abs().max())), err_msg=n)

    # conv -> BN -> (no ReLU) -> max pool -> ReLU: winners taken over the pre-ReLU values
    stem, c2c = "norelu/InceptionV1/Conv2d_1a_7x7", "norelu/InceptionV1/Conv2d_2c_3x3"
    probe = R.DeepSentimentRef(params, None, "image", torch.float64)
    probe.inject, probe.record = {stem: True}, {}
    probe.forward(batch)          # only the first pool's record is meaningful (everything behind it lacks a ReLU)
    first = {stem: True, "pool/MaxPool_2a_3x3": probe.record["pool/MaxPool_2a_3x3"],
             "poolrelu/MaxPool_2a_3x3": rec["poolrelu/MaxPool_2a_3x3"]}
    probe = R.DeepSentimentRef(params, None, "image", torch.float64)
    probe.inject, probe.record = dict(first, **{c2c: True}), {}
    probe.forward(batch)          # the stem is right now, so the second pool's pre-ReLU winners are too
    inj2 = R.DeepSentimentRef(params, None, "image", torch.float64)
    inj2.inject = dict(inj.inject, **first)
    inj2.inject.update({c2c: True, "pool/MaxPool_3a_3x3": probe.record["pool/MaxPool_3a_3x3"],
                        "poolrelu/MaxPool_3a_3x3": rec["poolrelu/MaxPool_3a_3x3"]})
    out3 = inj2.train_step(batch, 1e-3)
    np.testing.assert_allclose(out3["logits"].numpy(), out["logits"].numpy(), atol=1e-12)
    for n, g in out["grads"].items():
        np.testing.assert_allclose(out3["grads"][n].numpy(), g.numpy(), atol=1e-12 * max(1.0, float(g.abs().max())), err_msg=n)
