"""NumPy restatement of the TensorFlow-1.x semantics the Deep Sentiment path relies on.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY: pinned for SAME-conv, BN moving
stats, endpoint shapes and the variable count by the reference's own known answers
(tests/test_oracle_known_answers.py); **parity unpinned** for LSTM / head / CE / Adam.

Every function cites the reference call site (relative to /root/reference) whose behaviour it
restates.  All functions are dtype-generic: pass float64 arrays for a high-precision oracle,
float32 to mimic the reference's arithmetic type.
"""
import numpy as np

# ----------------------------------------------------------------------------------------------
# Hyper-parameters fixed by the reference
# ----------------------------------------------------------------------------------------------
WEIGHT_DECAY = 0.00004      # slim/nets/inception_utils.py:32
BN_DECAY = 0.9997           # slim/nets/inception_utils.py:34
BN_EPS = 0.001              # slim/nets/inception_utils.py:35
DROPOUT_KEEP = 0.8          # image_model/inception_v1.py:257
FORGET_BIAS = 1.0           # tf.contrib.rnn.BasicLSTMCell default, used at im_text_rnn_model.py:89
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8   # tf.train.AdamOptimizer defaults, :134


# ----------------------------------------------------------------------------------------------
# Topology of image_model/inception_v1.py (restated as a table, not copied code)
# ----------------------------------------------------------------------------------------------
# ("conv", name, k, stride, cout) | ("maxpool", name, k, stride) | ("mixed", name, b0,(b1a,b1b),(b2a,b2b),b3)
INCEPTION_V1 = [
    ("conv", "Conv2d_1a_7x7", 7, 2, 64),                       # inception_v1.py:62-63
    ("maxpool", "MaxPool_2a_3x3", 3, 2),                        # :66-67
    ("conv", "Conv2d_2b_1x1", 1, 1, 64),                        # :70-71
    ("conv", "Conv2d_2c_3x3", 3, 1, 192),                       # :74-75
    ("maxpool", "MaxPool_3a_3x3", 3, 2),                        # :78-79
    ("mixed", "Mixed_3b", 64, (96, 128), (16, 32), 32),         # :83-96
    ("mixed", "Mixed_3c", 128, (128, 192), (32, 96), 64),       # :100-113
    ("maxpool", "MaxPool_4a_3x3", 3, 2),                        # :117-118
    ("mixed", "Mixed_4b", 192, (96, 208), (16, 48), 64),        # :122-135
    ("mixed", "Mixed_4c", 160, (112, 224), (24, 64), 64),       # :139-152
    ("mixed", "Mixed_4d", 128, (128, 256), (24, 64), 64),       # :156-169
    ("mixed", "Mixed_4e", 112, (144, 288), (32, 64), 64),       # :173-186
    ("mixed", "Mixed_4f", 256, (160, 320), (32, 128), 128),     # :190-203
    ("maxpool", "MaxPool_5a_2x2", 2, 2),                        # :207-208
    ("mixed", "Mixed_5b", 256, (160, 320), (32, 128), 128),     # :212-225
    ("mixed", "Mixed_5c", 384, (192, 384), (48, 128), 128),     # :235-248  (trainable, :229-231)
]
TRAINABLE_BLOCKS = ("Mixed_5c",)   # inception_v1.py:229-231; everything before is trainable=False :57-59


def mixed_conv_names(block):
    """Scope names of the six convs of a Mixed block, in graph-construction order.
    Mixed_5b's Branch_2 3x3 is named Conv2d_0a_3x3 in the reference (inception_v1.py:221)."""
    b2b = "Conv2d_0a_3x3" if block == "Mixed_5b" else "Conv2d_0b_3x3"
    return [
        ("Branch_0/Conv2d_0a_1x1", 1, "in", 0),
        ("Branch_1/Conv2d_0a_1x1", 1, "in", 1),
        ("Branch_1/Conv2d_0b_3x3", 3, "b1a", 2),
        ("Branch_2/Conv2d_0a_1x1", 1, "in", 3),
        ("Branch_2/" + b2b, 3, "b2a", 4),
        ("Branch_3/Conv2d_0b_1x1", 1, "pool", 5),
    ]


def conv_layer_table(in_channels=3):
    """[(scope, k, stride, cin, cout, trainable)] for the 57 conv+BN layers, construction order."""
    out = []
    c = in_channels
    for item in INCEPTION_V1:
        if item[0] == "conv":
            _, name, k, s, co = item
            out.append(("InceptionV1/" + name, k, s, c, co, False))
            c = co
        elif item[0] == "mixed":
            _, name, b0, (b1a, b1b), (b2a, b2b), b3 = item
            tr = name in TRAINABLE_BLOCKS
            couts = [b0, b1a, b1b, b2a, b2b, b3]
            cins = {"in": c, "b1a": b1a, "b2a": b2a, "pool": c}
            for (scope, k, src, idx) in mixed_conv_names(name):
                out.append(("InceptionV1/%s/%s" % (name, scope), k, 1, cins[src], couts[idx], tr))
            c = b0 + b1b + b2b + b3
    return out


# ----------------------------------------------------------------------------------------------
# A1  SAME padding (TF): out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0);
#     before = pad_total // 2 (extra goes bottom/right).  Pinned by slim/nets/resnet_v1_test.py:72-152.
# ----------------------------------------------------------------------------------------------
def same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    before = total // 2
    return out, before, total - before


def _patches(x, kh, kw, s, pad_value, mode):
    """x [N,H,W,C] -> windows [N,OH,OW,kh,kw,C] with TF SAME or VALID geometry."""
    n, h, w, c = x.shape
    if mode == "SAME":
        oh, pt, pb = same_pad(h, kh, s)
        ow, pl, pr = same_pad(w, kw, s)
    else:
        oh, ow = (h - kh) // s + 1, (w - kw) // s + 1
        pt = pb = pl = pr = 0
    xp = np.full((n, h + pt + pb, w + pl + pr, c), pad_value, dtype=x.dtype)
    xp[:, pt:pt + h, pl:pl + w, :] = x
    sn, sh, sw, sc = xp.strides
    win = np.lib.stride_tricks.as_strided(
        xp, shape=(n, oh, ow, kh, kw, c), strides=(sn, sh * s, sw * s, sh, sw, sc), writeable=False)
    return win, (pt, pl), xp.shape


def conv2d_same(x, w, stride=1):
    """slim.conv2d core (no bias / BN / activation): NHWC x, HWIO w, cross-correlation,
    zero SAME padding.  image_model/inception_v1.py:60-63 (arg-scope padding='SAME')."""
    kh, kw, ci, co = w.shape
    win, _, _ = _patches(x, kh, kw, stride, 0.0, "SAME")
    return np.tensordot(win, w, axes=([3, 4, 5], [0, 1, 2]))


def conv2d_same_bwd_input(dy, w, x_shape, stride=1):
    """Conv2DBackpropInput for conv2d_same (implied by create_train_op, im_text_rnn_model.py:135)."""
    n, h, wd, c = x_shape
    kh, kw, ci, co = w.shape
    oh, pt, pb = same_pad(h, kh, stride)
    ow, pl, pr = same_pad(wd, kw, stride)
    dxp = np.zeros((n, h + pt + pb, wd + pl + pr, c), dtype=dy.dtype)
    for i in range(kh):
        for j in range(kw):
            contrib = np.tensordot(dy, w[i, j], axes=([3], [1]))      # [N,OH,OW,Ci]
            dxp[:, i:i + oh * stride:stride, j:j + ow * stride:stride, :] += contrib
    return dxp[:, pt:pt + h, pl:pl + wd, :]


def conv2d_same_bwd_filter(x, dy, w_shape, stride=1):
    """Conv2DBackpropFilter for conv2d_same."""
    kh, kw, ci, co = w_shape
    win, _, _ = _patches(x, kh, kw, stride, 0.0, "SAME")
    return np.tensordot(win, dy, axes=([0, 1, 2], [0, 1, 2]))


def max_pool(x, k, stride, mode="SAME"):
    """slim.max_pool2d; SAME padding cells are ignored (-inf).  inception_v1.py:67,79,94,118,208."""
    win, _, _ = _patches(x, k, k, stride, -np.inf, mode)
    return win.max(axis=(3, 4))


def max_pool_argmax(x, k, stride, mode="SAME"):
    """Winner of every window as the tap index kh*k+kw inside the (padded) window, first maximum in
    row-major order -- the element MaxPoolGrad routes the gradient to.  uint8 [N,OH,OW,C]."""
    n, h, w, c = x.shape
    win, _, _ = _patches(x, k, k, stride, -np.inf, mode)
    return win.reshape(n, win.shape[1], win.shape[2], k * k, c).argmax(axis=3).astype(np.uint8)


def max_pool_bwd(x, dy, k, stride, mode="SAME"):
    """MaxPoolGrad: the gradient of each window goes to its first (row-major) maximal element."""
    n, h, w, c = x.shape
    win, (pt, pl), pshape = _patches(x, k, k, stride, -np.inf, mode)
    oh, ow = win.shape[1], win.shape[2]
    flat = win.reshape(n, oh, ow, k * k, c)
    arg = flat.argmax(axis=3)                                    # first max, row-major
    dxp = np.zeros(pshape, dtype=dy.dtype)
    for t in range(k * k):
        i, j = divmod(t, k)
        dxp[:, i:i + oh * stride:stride, j:j + ow * stride:stride, :] += np.where(arg == t, dy, 0)
    return dxp[:, pt:pt + h, pl:pl + w, :]


def avg_pool_valid(x, k):
    """slim.avg_pool2d(net,[7,7],stride=1) default VALID.  inception_v1.py:299."""
    win, _, _ = _patches(x, k, k, 1, 0.0, "VALID")
    return win.mean(axis=(3, 4))


# ----------------------------------------------------------------------------------------------
# A3  slim.batch_norm(center=True, scale=False), train mode.  slim/nets/inception_utils.py:48-70.
#     Pinned (moving statistics) by slim/deployment/model_deploy_test.py:467-524.
# ----------------------------------------------------------------------------------------------
def batch_norm_train(z, beta, eps=BN_EPS):
    axes = tuple(range(z.ndim - 1))
    mean = z.mean(axis=axes)
    var = ((z - mean) ** 2).mean(axis=axes)          # biased
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (z - mean) * rstd
    return xhat + beta, mean, var, xhat, rstd


def batch_norm_moving_update(mm, mv, mean, var, decay=BN_DECAY):
    """assign_moving_average: m <- decay*m + (1-decay)*batch."""
    return decay * mm + (1 - decay) * mean, decay * mv + (1 - decay) * var


def batch_norm_infer(z, beta, mm, mv, eps=BN_EPS):
    return (z - mm) / np.sqrt(mv + eps) + beta


def batch_norm_train_bwd(dy, xhat, rstd):
    """No gamma: dbeta = sum(dy); dz = rstd * (dy - mean(dy) - xhat * mean(dy*xhat))."""
    axes = tuple(range(dy.ndim - 1))
    dbeta = dy.sum(axis=axes)
    dz = rstd * (dy - dy.mean(axis=axes) - xhat * (dy * xhat).mean(axis=axes))
    return dz, dbeta


def relu(x):
    return np.maximum(x, 0)


def conv_bn_relu(x, w, beta, stride=1, eps=BN_EPS):
    """One slim.conv2d under inception_arg_scope: conv -> BN(train, beta only) -> ReLU."""
    z = conv2d_same(x, w, stride)
    y, mean, var, xhat, rstd = batch_norm_train(z, beta, eps)
    return relu(y), dict(z=z, mean=mean, var=var, xhat=xhat, rstd=rstd)


def dropout(x, keep, mask):
    """slim.dropout train mode with an injected Bernoulli(keep) mask.  inception_v1.py:300-301."""
    return x * mask / keep


# ----------------------------------------------------------------------------------------------
# Inception-v1 forward (train-mode BN), image_model/inception_v1.py:29-309
# ----------------------------------------------------------------------------------------------
def inception_v1_forward(images, params, num_classes_key="InceptionV1/Logits/Conv2d_0c_1x1",
                         dropout_mask=None, keep=DROPOUT_KEEP, final_endpoint="Mixed_5c"):
    """images NHWC in [-1,1]; params: dict of TF-named arrays (weights HWIO, BatchNorm/beta,
    Logits weights/biases).  Returns (logits [N,num_classes], end_points)."""
    ep = {}
    bn_stats = {}
    net = images

    def cbr(x, scope, stride=1):
        y, aux = conv_bn_relu(x, params[scope + "/weights"], params[scope + "/BatchNorm/beta"], stride)
        bn_stats[scope] = (aux["mean"], aux["var"])
        return y

    for item in INCEPTION_V1:
        kind, name = item[0], item[1]
        if kind == "conv":
            net = cbr(net, "InceptionV1/" + name, item[3])
        elif kind == "maxpool":
            net = max_pool(net, item[2], item[3], "SAME")
        else:
            pre = "InceptionV1/%s/" % name
            names = [n for (n, _, _, _) in mixed_conv_names(name)]
            b0 = cbr(net, pre + names[0])
            b1 = cbr(cbr(net, pre + names[1]), pre + names[2])
            b2 = cbr(cbr(net, pre + names[3]), pre + names[4])
            b3 = cbr(max_pool(net, 3, 1, "SAME"), pre + names[5])
            net = np.concatenate([b0, b1, b2, b3], axis=3)
        ep[name] = net
        if name == final_endpoint:
            break
    pooled = avg_pool_valid(net, 7)                       # [N,1,1,C] for 224x224 inputs
    assert pooled.shape[1] == 1 and pooled.shape[2] == 1, "SpatialSqueeze needs a 1x1 map (:305)"
    pooled = pooled[:, 0, 0, :]
    ep["AvgPool_0a_7x7"] = pooled
    if dropout_mask is not None:
        pooled = dropout(pooled, keep, dropout_mask)
    w = params[num_classes_key + "/weights"]
    logits = pooled @ w.reshape(w.shape[2], w.shape[3]) + params[num_classes_key + "/biases"]
    ep["Logits"] = logits
    ep["_bn_stats"] = bn_stats
    return logits, ep


# ----------------------------------------------------------------------------------------------
# A7/A8  BasicLSTMCell + dynamic_rnn(sequence_length) + gather_nd(last valid step)
#        image_text_model/im_text_rnn_model.py:85-92 ; text_model/text_embedding.py:75-82
# ----------------------------------------------------------------------------------------------
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def embedding_lookup(table, ids):
    """tf.nn.embedding_lookup(W_embedding, texts).  im_text_rnn_model.py:85."""
    return table[ids]


def lstm_forward(x, seq_len, kernel, bias, forget_bias=FORGET_BIAS, keep_cache=False, initial_state=None):
    """x [B,T,D]; kernel [D+H,4H] applied to concat([x_t,h]); gate order i,j,f,o.
    For t >= seq_len[b] the output row is zero and (c,h) are carried.  Returns
    (outputs [B,T,H], h_last [B,H] = outputs[b, seq_len[b]-1])."""
    b, t, d = x.shape
    hsz = kernel.shape[1] // 4
    c = np.zeros((b, hsz), dtype=x.dtype)
    h = np.zeros((b, hsz), dtype=x.dtype)
    if initial_state is not None:          # (c, h); the reference always starts from zeros (dynamic_rnn default)
        c, h = (np.asarray(a, dtype=x.dtype) for a in initial_state)
    outs = np.zeros((b, t, hsz), dtype=x.dtype)
    cache = []
    for s in range(t):
        z = np.concatenate([x[:, s, :], h], axis=1) @ kernel + bias
        i, j, f, o = np.split(z, 4, axis=1)
        si, sf, so, tj = sigmoid(i), sigmoid(f + forget_bias), sigmoid(o), np.tanh(j)
        c_new = c * sf + si * tj
        tc = np.tanh(c_new)
        h_new = tc * so
        live = (s < seq_len)[:, None]
        if keep_cache:
            cache.append(dict(xh=np.concatenate([x[:, s, :], h], axis=1), c_prev=c, si=si, sf=sf, so=so,
                              tj=tj, tc=tc, live=live))
        outs[:, s, :] = np.where(live, h_new, 0)
        c = np.where(live, c_new, c)
        h = np.where(live, h_new, h)
    h_last = outs[np.arange(b), seq_len - 1]
    if keep_cache:
        return outs, h_last, cache
    return outs, h_last


def lstm_backward(dh_last, seq_len, kernel, cache, return_dz=False):
    """BPTT for lstm_forward given d(loss)/d(h_last).  Returns (dkernel, dbias).  The embedding
    is frozen (trainable=False, im_text_rnn_model.py:82) so no dx is produced.  return_dz: also the
    per-step gradient of the gate pre-activations [T][B,4H] (zero rows past seq_len)."""
    d_in = cache[0]["xh"].shape[1] - dh_last.shape[1]
    dk = np.zeros_like(kernel)
    db = np.zeros(kernel.shape[1], dtype=kernel.dtype)
    dh = np.zeros_like(dh_last)
    dc = np.zeros_like(dh_last)
    dzs = [None] * len(cache)
    for s in reversed(range(len(cache))):
        q = cache[s]
        last = (s == seq_len - 1)[:, None]
        dh = dh + np.where(last, dh_last, 0)
        live = q["live"]
        do = dh * q["tc"] * q["so"] * (1 - q["so"])
        dct = dc + dh * q["so"] * (1 - q["tc"] ** 2)
        di = dct * q["tj"] * q["si"] * (1 - q["si"])
        dj = dct * q["si"] * (1 - q["tj"] ** 2)
        df = dct * q["c_prev"] * q["sf"] * (1 - q["sf"])
        dz = np.where(live, np.concatenate([di, dj, df, do], axis=1), 0)
        dzs[s] = dz
        dk += q["xh"].T @ dz
        db += dz.sum(axis=0)
        dxh = dz @ kernel.T
        dh = np.where(live, dxh[:, d_in:], dh)
        dc = np.where(live, dct * q["sf"], dc)
    if return_dz:
        return dk, db, dzs
    return dk, db


# ----------------------------------------------------------------------------------------------
# A9  loss: slim.losses.softmax_cross_entropy (mean over batch) + L2 of every conv `weights`
#     image_text_model/im_text_rnn_model.py:124-126 ; slim/nets/inception_utils.py:63-64
# ----------------------------------------------------------------------------------------------
def softmax_cross_entropy(logits, labels):
    m = logits.max(axis=1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(logits - m).sum(axis=1))
    ce = lse - logits[np.arange(logits.shape[0]), labels]
    return ce.mean()


def softmax_cross_entropy_grad(logits, labels):
    m = logits.max(axis=1, keepdims=True)
    e = np.exp(logits - m)
    p = e / e.sum(axis=1, keepdims=True)
    p[np.arange(logits.shape[0]), labels] -= 1
    return p / logits.shape[0]


def l2_regularizer(w, scale=WEIGHT_DECAY):
    """slim.l2_regularizer(scale)(w) = scale * sum(w**2) / 2."""
    return scale * 0.5 * np.sum(np.square(w))


# ----------------------------------------------------------------------------------------------
# A10  tf.train.AdamOptimizer (epsilon outside the bias correction)
# ----------------------------------------------------------------------------------------------
def adam_step(w, g, m, v, t, lr, b1=ADAM_B1, b2=ADAM_B2, eps=ADAM_EPS):
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    w = w - lr_t * m / (np.sqrt(v) + eps)
    return w, m, v


# ----------------------------------------------------------------------------------------------
# A6  initialisers
# ----------------------------------------------------------------------------------------------
def trunc_normal(rng, shape, stddev):
    """tf.truncated_normal_initializer: N(0, stddev^2) re-drawn outside +-2 stddev."""
    out = rng.standard_normal(size=shape)
    bad = np.abs(out) > 2
    while bad.any():
        out[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(out) > 2
    return out * stddev


def variance_scaling(rng, shape):
    """slim.variance_scaling_initializer() defaults: factor 2.0, FAN_IN, truncated normal with
    stddev sqrt(1.3 * 2 / fan_in).  Used by the Logits conv (inception_utils.py:67)."""
    fan_in = int(np.prod(shape[:-1]))
    return trunc_normal(rng, shape, np.sqrt(1.3 * 2.0 / fan_in))


def glorot_uniform(rng, shape):
    """tf.get_variable default initialiser.  For a 1-D shape [n] fan_in = fan_out = n."""
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    else:
        fan_in, fan_out = int(np.prod(shape[:-1])), shape[-1]
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape)


def init_inception_params(rng, num_classes, dtype=np.float32):
    """Variables slim creates for inception_v1(..., num_classes): per conv `weights`
    [k,k,cin,cout] + BatchNorm/{beta,moving_mean,moving_variance}; Logits weights+biases."""
    p = {}
    for (scope, k, s, ci, co, tr) in conv_layer_table():
        p[scope + "/weights"] = trunc_normal(rng, (k, k, ci, co), 0.01).astype(dtype)
        p[scope + "/BatchNorm/beta"] = np.zeros(co, dtype)
        p[scope + "/BatchNorm/moving_mean"] = np.zeros(co, dtype)
        p[scope + "/BatchNorm/moving_variance"] = np.ones(co, dtype)
    p["InceptionV1/Logits/Conv2d_0c_1x1/weights"] = variance_scaling(rng, (1, 1, 1024, num_classes)).astype(dtype)
    p["InceptionV1/Logits/Conv2d_0c_1x1/biases"] = np.zeros(num_classes, dtype)
    return p


def init_text_params(rng, embed_dim, rnn_size, dtype=np.float32):
    return {
        "Text/rnn/basic_lstm_cell/kernel": glorot_uniform(rng, (embed_dim + rnn_size, 4 * rnn_size)).astype(dtype),
        "Text/rnn/basic_lstm_cell/bias": np.zeros(4 * rnn_size, dtype),
    }


def init_joint_head(rng, in_size, fc_size, nb_emotions, dtype=np.float32):
    """W_fc, b_fc, W_softmax, b_softmax: all tf.get_variable without initializer -> glorot
    uniform, biases included.  image_text_model/im_text_rnn_model.py:98-104."""
    return {
        "W_fc": glorot_uniform(rng, (in_size, fc_size)).astype(dtype),
        "b_fc": glorot_uniform(rng, (fc_size,)).astype(dtype),
        "W_softmax": glorot_uniform(rng, (fc_size, nb_emotions)).astype(dtype),
        "b_softmax": glorot_uniform(rng, (nb_emotions,)).astype(dtype),
    }


def init_text_head(rng, rnn_size, nb_emotions, dtype=np.float32):
    """text_model/text_embedding.py:84-85."""
    return {
        "W_softmax": glorot_uniform(rng, (rnn_size, nb_emotions)).astype(dtype),
        "b_softmax": glorot_uniform(rng, (nb_emotions,)).astype(dtype),
    }


# ----------------------------------------------------------------------------------------------
# Synthetic batches
# ----------------------------------------------------------------------------------------------
def synthetic_batch(batch, post_size, vocab, nb_emotions=15, image_size=224, seed=0, with_images=True):
    rng = np.random.RandomState(seed)
    out = {}
    if with_images:
        out["images"] = rng.uniform(-1, 1, size=(batch, image_size, image_size, 3)).astype(np.float32)
    seq_len = rng.randint(min(6, post_size), post_size + 1, size=batch).astype(np.int64)
    ids = rng.randint(0, vocab, size=(batch, post_size)).astype(np.int64)
    ids[np.arange(post_size)[None, :] >= seq_len[:, None]] = vocab      # pad id = unk id = V
    out["texts"] = ids
    out["seq_lens"] = seq_len
    out["labels"] = rng.randint(0, nb_emotions, size=batch).astype(np.int64)
    return out


def synthetic_embedding(vocab, dim, seed=1):
    rng = np.random.RandomState(seed)
    emb = rng.normal(0, 0.4, size=(vocab + 1, dim)).astype(np.float32)
    emb[vocab] = 0            # '<ukn>' / pad row is zeros, im_text_rnn_model.py:75
    return emb


# ----------------------------------------------------------------------------------------------
# Joint / text-only forward (NumPy), image_text_model/im_text_rnn_model.py:38-105
# ----------------------------------------------------------------------------------------------
def text_tower_forward(params, embedding, texts, seq_lens):
    x = embedding_lookup(embedding, texts)
    _, h_last = lstm_forward(x, seq_lens, params["Text/rnn/basic_lstm_cell/kernel"],
                             params["Text/rnn/basic_lstm_cell/bias"])
    return h_last


def deep_sentiment_forward(params, embedding, images, texts, seq_lens, dropout_mask=None):
    im_feat, ep = inception_v1_forward(images, params, dropout_mask=dropout_mask)
    tx_feat = text_tower_forward(params, embedding, texts, seq_lens)
    concat = np.concatenate([im_feat, tx_feat], axis=1)                    # :95
    dense = relu(concat @ params["W_fc"] + params["b_fc"])                 # :98-101
    logits = dense @ params["W_softmax"] + params["b_softmax"]             # :103-105
    return logits, dict(im_feat=im_feat, tx_feat=tx_feat, concat=concat, end_points=ep)


def total_loss(logits, labels, params, with_l2=True):
    """CE + sum of L2 over every conv `weights` (frozen included).  :124-126."""
    loss = softmax_cross_entropy(logits, labels)
    if with_l2:
        for name, w in params.items():
            if name.startswith("InceptionV1/") and name.endswith("/weights"):
                loss = loss + l2_regularizer(w)
    return loss
