"""Pin the oracle against every numeric known answer the reference tree holds for the path.

(1) slim/nets/resnet_v1_test.py:72-152   slim.conv2d SAME padding, stride 1 and 2, even and odd
(2) slim/deployment/model_deploy_test.py:467-524   BatchNorm moving mean / variance
(3) slim/nets/inception_v1_test.py:85-100, 119-127   endpoint shapes, half-size input
(4) slim/nets/inception_v1_test.py:109-117   5 607 184 model variables
"""
import numpy as np
import pytest
import torch

from oracle import tf_semantics as S
from oracle import torch_ref as R


def _mesh(n):
    # create_test_input(1, n, n, 1): value = row + col   (resnet_v1_test.py:47-55)
    return (np.arange(n)[:, None] + np.arange(n)[None, :]).astype(np.float64).reshape(1, n, n, 1)


def _both_convs(x, w, stride):
    y_np = S.conv2d_same(x, w, stride)[0, :, :, 0]
    y_t = R.conv2d_same(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(w), stride)[0, 0].numpy()
    return y_np, y_t


def test_conv2d_same_even_known_answer():
    x = _mesh(4)
    w = _mesh(3).reshape(3, 3, 1, 1)
    y1 = [[14, 28, 43, 26], [28, 48, 66, 37], [43, 66, 84, 46], [26, 37, 46, 22]]
    y4 = [[48, 37], [37, 22]]          # stride 2: padding goes bottom/right
    for got in _both_convs(x, w, 1):
        np.testing.assert_allclose(got, y1)
    for got in _both_convs(x, w, 2):
        np.testing.assert_allclose(got, y4)


def test_conv2d_same_odd_known_answer():
    x = _mesh(5)
    w = _mesh(3).reshape(3, 3, 1, 1)
    y1 = [[14, 28, 43, 58, 34], [28, 48, 66, 84, 46], [43, 66, 84, 102, 55],
          [58, 84, 102, 120, 64], [34, 46, 55, 64, 30]]
    y4 = [[14, 43, 34], [43, 84, 55], [34, 55, 30]]
    for got in _both_convs(x, w, 1):
        np.testing.assert_allclose(got, y1)
    for got in _both_convs(x, w, 2):
        np.testing.assert_allclose(got, y4)


def test_batch_norm_moving_stats_known_answer():
    # DeployTest.setUp (model_deploy_test.py:467-477) + BatchNormClassifier decay=0.1 (:161)
    np.random.seed(0)
    inputs = np.zeros((16, 4))
    labels = np.random.randint(0, 2, size=(16, 1)).astype(np.float32)
    for i in range(16):
        j = int(2 * labels[i, 0] + np.random.randint(0, 2))
        inputs[i, j] = 1
    mm, mv = np.zeros(4), np.ones(4)
    for _ in range(10):
        _, mean, var, _, _ = S.batch_norm_train(inputs, np.zeros(4))
        mm, mv = S.batch_norm_moving_update(mm, mv, mean, var, decay=0.1)
    np.testing.assert_allclose(mm, [0.125, 0.25, 0.375, 0.25], rtol=1e-6)
    np.testing.assert_allclose(mv, [0.109375, 0.1875, 0.234375, 0.1875], rtol=1e-6)
    # torch restatement agrees on the batch statistics
    z = torch.tensor(inputs).reshape(16, 4, 1, 1)
    _, tmean, tvar = R.batch_norm_train(z, torch.zeros(4, dtype=torch.float64))
    np.testing.assert_allclose(tmean.numpy(), mean, rtol=1e-12)
    np.testing.assert_allclose(tvar.numpy(), var, rtol=1e-12)


def test_variable_count_5607184():
    n = 0
    for (scope, k, s, ci, co, tr) in S.conv_layer_table():
        n += k * k * ci * co + 3 * co          # weights + beta + moving_mean + moving_variance
    assert n == 5607184
    table = S.conv_layer_table()
    assert len(table) == 57
    assert sum(k * k * ci * co for (_, k, _, ci, co, _) in table) == 5585344
    assert sum(co for (_, _, _, _, co, _) in table) == 7280
    assert sum(k * k * ci * co for (_, k, _, ci, co, tr) in table if tr) == 1344512


ENDPOINT_SHAPES = {'Conv2d_1a_7x7': [112, 112, 64], 'MaxPool_2a_3x3': [56, 56, 64],
                   'Conv2d_2b_1x1': [56, 56, 64], 'Conv2d_2c_3x3': [56, 56, 192],
                   'MaxPool_3a_3x3': [28, 28, 192], 'Mixed_3b': [28, 28, 256], 'Mixed_3c': [28, 28, 480],
                   'MaxPool_4a_3x3': [14, 14, 480], 'Mixed_4b': [14, 14, 512], 'Mixed_4c': [14, 14, 512],
                   'Mixed_4d': [14, 14, 512], 'Mixed_4e': [14, 14, 528], 'Mixed_4f': [14, 14, 832],
                   'MaxPool_5a_2x2': [7, 7, 832], 'Mixed_5b': [7, 7, 832], 'Mixed_5c': [7, 7, 1024]}


def test_endpoint_shapes_full_and_half_size():
    rng = np.random.RandomState(0)
    params = S.init_inception_params(rng, 15)
    x = rng.uniform(-1, 1, size=(1, 224, 224, 3)).astype(np.float32)
    logits, ep = S.inception_v1_forward(x, params)
    for name, shp in ENDPOINT_SHAPES.items():
        assert list(ep[name].shape) == [1] + shp, name
    assert logits.shape == (1, 15)
    # half-size images -> Mixed_5c [4,4,1024]  (inception_v1_test.py:119-127)
    net = x[:, :112, :112, :]
    for item in S.INCEPTION_V1:
        if item[0] == "conv":
            h = S.same_pad(net.shape[1], item[2], item[3])[0]
            net = np.zeros((1, h, h, item[4]), np.float32)
        elif item[0] == "maxpool":
            h = S.same_pad(net.shape[1], item[2], item[3])[0]
            net = np.zeros((1, h, h, net.shape[3]), np.float32)
        else:
            net = np.zeros((1, net.shape[1], net.shape[2], item[2] + item[3][1] + item[4][1] + item[5]), np.float32)
    assert net.shape == (1, 4, 4, 1024)


def test_numpy_vs_torch_inception_forward_fp64():
    rng = np.random.RandomState(3)
    params = S.init_inception_params(rng, 7, dtype=np.float64)
    # non-trivial betas so the BN shift is exercised
    for k in params:
        if k.endswith("beta"):
            params[k] = rng.normal(0, 0.1, size=params[k].shape)
    x = rng.uniform(-1, 1, size=(2, 224, 224, 3))
    mask = (rng.uniform(size=(2, 1024)) < 0.8).astype(np.float64)
    logits_np, ep = S.inception_v1_forward(x, params, dropout_mask=mask)
    ref = R.DeepSentimentRef(params, mode="image", dtype=torch.float64)
    logits_t = ref.forward(dict(images=x), torch.tensor(mask)).detach().numpy()
    np.testing.assert_allclose(logits_np, logits_t, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(ep["Mixed_5c"], ref.last_mixed_5c.detach().permute(0, 2, 3, 1).numpy(),
                               rtol=1e-8, atol=1e-10)


# ------------------------------------------------------------------------------------------------------
# [TF-sem] BasicLSTMCell known answer.  NOT under /root/reference (TensorFlow is a third-party dependency
# of the reference and is not vendored): recalled from TensorFlow 1.x's own unit test
# tensorflow/python/kernel_tests/rnn_cell_test.py::testBasicLSTMCell -- a 2-layer stack of
# BasicLSTMCell(2), every kernel entry 0.5, zero bias, x = [1, 1], every state 0.1; that test asserts the
# state vector [c1, c1, h1, h1, c2, c2, h2, h2] below with assertAllClose (rtol 1e-6; TF computed in fp32).
# It pins gate order i,j,f,o, forget_bias = 1.0 added before the sigmoid, tanh, and
# h' = tanh(c') * sigmoid(o) for the restatements of im_text_rnn_model.py:89.
# ------------------------------------------------------------------------------------------------------
TF_BASIC_LSTM_EXPECTED = [0.68967271, 0.68967271, 0.44848421, 0.44848421,
                          0.39897051, 0.39897051, 0.24024698, 0.24024698]


def test_basic_lstm_cell_tf_known_answer_numpy():
    k = np.full((4, 8), 0.5)
    b = np.zeros(8)
    st = (np.full((1, 2), 0.1), np.full((1, 2), 0.1))
    seq = np.array([1])
    _, h1, cache1 = S.lstm_forward(np.array([[[1.0, 1.0]]]), seq, k, b, keep_cache=True, initial_state=st)
    c1 = cache1[0]["c_prev"] * cache1[0]["sf"] + cache1[0]["si"] * cache1[0]["tj"]
    _, h2, cache2 = S.lstm_forward(h1[:, None, :], seq, k, b, keep_cache=True, initial_state=st)
    c2 = cache2[0]["c_prev"] * cache2[0]["sf"] + cache2[0]["si"] * cache2[0]["tj"]
    np.testing.assert_allclose(np.concatenate([c1, h1, c2, h2], axis=1)[0], TF_BASIC_LSTM_EXPECTED, rtol=1e-6)
    np.testing.assert_allclose(np.arctanh(h1 / cache1[0]["so"]), c1, rtol=1e-12)      # h' = tanh(c') * sigmoid(o)


def test_basic_lstm_cell_tf_known_answer_torch_ref():
    k = np.full((4, 8), 0.5)
    ref = R.DeepSentimentRef({"Text/rnn/basic_lstm_cell/kernel": k, "Text/rnn/basic_lstm_cell/bias": np.zeros(8),
                              "W_softmax": np.zeros((2, 2)), "b_softmax": np.zeros(2)},
                             embedding=np.array([[1.0, 1.0]]), mode="text", dtype=torch.float64)
    st = (np.full((1, 2), 0.1), np.full((1, 2), 0.1))
    ids, seq = torch.zeros(1, 1, dtype=torch.int64), torch.tensor([1])
    h1 = ref.text_tower(ids, seq, initial_state=st).detach()
    ref.embedding = h1.clone()                        # layer 2 reads layer 1's output
    h2 = ref.text_tower(ids, seq, initial_state=st).detach()
    np.testing.assert_allclose(torch.cat([h1, h2], dim=1)[0].numpy(),
                               [TF_BASIC_LSTM_EXPECTED[i] for i in (2, 3, 6, 7)], rtol=1e-6)


def test_third_lstm_restatement_torch_nn_lstm_agrees():
    """An independently authored LSTM -- torch.nn.LSTM (ATen's fused cell, gate order i,f,g,o, separate
    W_ih / W_hh and two biases) with pack_padded_sequence for dynamic_rnn's length masking -- against both
    restatements: forward (h at the last valid step) and the gradients of kernel and bias."""
    rng = np.random.RandomState(17)
    b, t, d, hsz = 6, 9, 5, 7
    x = rng.normal(size=(b, t, d))
    seq = np.array([9, 1, 4, 6, 2, 9])
    kernel = rng.normal(0, 0.4, size=(d + hsz, 4 * hsz))
    bias = rng.normal(0, 0.2, size=4 * hsz)
    dh = rng.normal(size=(b, hsz))

    # TF kernel columns are [i | j | f | o]; torch rows are [i | f | g(=j) | o]; forget_bias joins the f bias
    def remap(a):            # [..., 4H] in TF order -> torch order
        i, j, f, o = np.split(a, 4, axis=-1)
        return np.concatenate([i, f, j, o], axis=-1)
    lstm = torch.nn.LSTM(d, hsz, batch_first=True).double()
    fb = np.concatenate([np.zeros(hsz), np.full(hsz, S.FORGET_BIAS), np.zeros(2 * hsz)])
    with torch.no_grad():
        lstm.weight_ih_l0.copy_(torch.tensor(remap(kernel[:d]).T))
        lstm.weight_hh_l0.copy_(torch.tensor(remap(kernel[d:]).T))
        lstm.bias_ih_l0.copy_(torch.tensor(remap(bias) + fb))
        lstm.bias_hh_l0.zero_()
    packed = torch.nn.utils.rnn.pack_padded_sequence(torch.tensor(x), torch.tensor(seq), batch_first=True,
                                                     enforce_sorted=False)
    out, (h_n, _) = lstm(packed)
    h_last_nn = h_n[0]
    h_last_nn.backward(torch.tensor(dh))
    padded, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=t)

    outs, h_last, cache = S.lstm_forward(x, seq, kernel, bias, keep_cache=True)
    np.testing.assert_allclose(h_last_nn.detach().numpy(), h_last, atol=1e-13)
    np.testing.assert_allclose(padded.detach().numpy(), outs, atol=1e-13)          # zero rows past seq_len (A8)
    dk, db = S.lstm_backward(dh, seq, kernel, cache)

    def unmap(a):            # torch order -> TF order
        i, f, g, o = np.split(a, 4, axis=-1)
        return np.concatenate([i, g, f, o], axis=-1)
    dk_nn = np.concatenate([unmap(lstm.weight_ih_l0.grad.numpy().T), unmap(lstm.weight_hh_l0.grad.numpy().T)], axis=0)
    np.testing.assert_allclose(dk_nn, dk, atol=1e-12)
    np.testing.assert_allclose(unmap(lstm.bias_ih_l0.grad.numpy()), db, atol=1e-12)

    ref = R.DeepSentimentRef({"Text/rnn/basic_lstm_cell/kernel": kernel, "Text/rnn/basic_lstm_cell/bias": bias,
                              "W_softmax": np.zeros((hsz, 2)), "b_softmax": np.zeros(2)},
                             embedding=np.zeros((1, d)), mode="text", dtype=torch.float64)
    ref.embedding = torch.tensor(x.reshape(b * t, d))
    ht = ref.text_tower(torch.arange(b * t).reshape(b, t), torch.tensor(seq))
    ht.backward(torch.tensor(dh))
    np.testing.assert_allclose(ht.detach().numpy(), h_last_nn.detach().numpy(), atol=1e-13)
    np.testing.assert_allclose(ref.p["Text/rnn/basic_lstm_cell/kernel"].grad.numpy(), dk_nn, atol=1e-12)
