"""Inception-v1 front end with the reference's call signature (image_model/inception_v1.py:254-312).

`inception_v1(inputs, final_endpoint, num_classes, is_training, dropout_keep_prob, ...)` returns
(logits, end_points) like the reference; the graph-building body is replaced by
`InceptionV1Engine` (engine_image.py), i.e. HIP kernels.  Variables are created once per `scope`
(slim's variable_scope/reuse behaviour) and can be loaded from TF-named arrays.
"""
import torch

from ..engine_image import ENDPOINTS, TOPOLOGY, BN_DECAY, BN_EPS, WEIGHT_DECAY  # noqa: F401
from ..net import SentimentNet

default_image_size = 224          # image_model/inception_v1.py:310
_SCOPES = {}


def inception_v1_arg_scope(weight_decay=0.00004, use_batch_norm=True, batch_norm_decay=0.9997,
                           batch_norm_epsilon=0.001):
    """slim/nets/inception_utils.py:32-71.  The hyper-parameters are compiled into the engine; this
    returns them (and rejects values the kernels do not implement) so reference-style call sites keep working."""
    if (weight_decay, use_batch_norm, batch_norm_decay, batch_norm_epsilon) != (WEIGHT_DECAY, True, BN_DECAY, BN_EPS):
        raise NotImplementedError("only the reference's inception_arg_scope defaults are implemented")
    return dict(weight_decay=weight_decay, batch_norm_decay=batch_norm_decay, batch_norm_epsilon=batch_norm_epsilon)


def get_net(scope="InceptionV1", num_classes=1000, dropout_keep_prob=0.8, reuse=None, **kw):
    key = (scope, num_classes)
    if key not in _SCOPES or reuse is False:
        net = SentimentNet(mode="image", nb_emotions=num_classes, dropout_keep_prob=dropout_keep_prob, **kw)
        net.initialize()
        _SCOPES[key] = net
    return _SCOPES[key]


def inception_v1_base(inputs, final_endpoint="Mixed_5c", scope="InceptionV1", net=None):
    """Returns (activations at final_endpoint, end_points) -- image_model/inception_v1.py:29-251."""
    if final_endpoint not in ENDPOINTS:
        raise ValueError("Unknown final endpoint %s" % final_endpoint)      # :251
    net = net or get_net(scope)
    eng = net.image
    eng.fuse_bn_pool = False             # callers get every end point materialised
    if eng.zcat:                         # ... as activations, not as the pre-BatchNorm values a zcat concat keeps
        eng.zcat, eng.B = False, None
    with torch.no_grad():
        eng.forward(inputs, None, 0)
    end_points = {}
    for st in eng.stages:
        end_points[st.name] = st.out
        if st.name == final_endpoint:
            break
    return end_points[final_endpoint], end_points


def inception_v1(inputs, final_endpoint="Mixed_5c", num_classes=1000, is_training=True, dropout_keep_prob=0.8,
                 prediction_fn=None, spatial_squeeze=True, reuse=None, scope="InceptionV1", net=None,
                 dropout_mask=None):
    """Returns (logits [B,num_classes], end_points).  is_training=True: batch statistics + dropout (and, as in
    slim, the moving averages are updated by the training step, not by this call when no step follows);
    is_training=False: BatchNorm on the moving statistics, dropout off (image_model/inception_v1.py:295-296)."""
    if final_endpoint != "Mixed_5c":
        raise NotImplementedError("the Logits head sits on Mixed_5c (7x7 map); use inception_v1_base for earlier endpoints")
    if not spatial_squeeze:
        raise NotImplementedError("spatial_squeeze=False")
    net = net or get_net(scope, num_classes, dropout_keep_prob, reuse)
    net.image.fuse_bn_pool = False       # callers get every end point materialised
    if net.image.zcat:
        net.image.zcat, net.image.B = False, None
    if is_training:
        logits = net.forward({"images": inputs}, dropout_mask)
    else:
        logits = net.predict({"images": inputs}, is_training=False)
    end_points = {st.name: st.out for st in net.image.stages}
    end_points["Logits"] = logits
    return logits, end_points
