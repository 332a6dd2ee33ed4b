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
