"""Deep Sentiment (joint image + text) model and trainer with the reference's signatures
(image_text_model/im_text_rnn_model.py:38-169)."""
import numpy as np
import torch

from .. import ops
from ..image_model.im_model import get_init_fn
from ..net import SentimentNet
from ..training import SyntheticInput, run_training

_POST_SIZE = 50
_CONFIG = {'mode': 'train',
           'dataset_dir': 'data',
           'text_dir': 'text_model',
           'emb_dir': 'embedding_weights',
           'filename': 'glove.6B.50d.txt',
           'initial_lr': 1e-3,
           'decay_factor': 0.3,
           'batch_size': 64,
           'im_features_size': 256,
           'rnn_size': 1024,
           'final_endpoint': 'Mixed_5c',
           'fc_size': 512}                      # keys verbatim from im_text_rnn_model.py:24-35


class DeepSentiment(SyntheticInput):
    def __init__(self, config, nb_emotions=15, embedding=None, device="cuda", **net_kw):
        self.config = config
        if config.get('final_endpoint', 'Mixed_5c') != 'Mixed_5c':
            raise NotImplementedError("final_endpoint must be Mixed_5c")
        self.learning_rate = config['initial_lr']
        vocab = config.get('vocab_size', 10000)
        dim = config.get('embedding_dim', 50)
        post = config.get('post_size', _POST_SIZE)
        if embedding is not None:               # GloVe [V, D] + zero <ukn> row (:75-76)
            embedding = np.concatenate([np.asarray(embedding, np.float32),
                                        np.zeros((1, embedding.shape[1]), np.float32)])
            vocab, dim = embedding.shape[0] - 1, embedding.shape[1]
        self._init_input(config, post, vocab, nb_emotions, True, device)
        self.nb_emotions = self.dataset.num_classes
        for key in ("train_all", "trainable_embedding"):      # optional fine-tuning switches (not in the reference _CONFIG)
            if key in config:
                net_kw.setdefault(key, bool(config[key]))
        self.net = SentimentNet(mode="joint", nb_emotions=self.nb_emotions,
                                im_features_size=config['im_features_size'], rnn_size=config['rnn_size'],
                                fc_size=config['fc_size'], vocab_size=vocab, embedding_dim=dim, post_size=post,
                                device=device, **net_kw)
        self.net.initialize(seed=config.get('seed', 1))
        if embedding is not None:
            self.net.store.view("Text/W_embedding").copy_(torch.from_numpy(embedding))
        self.embedding = self.net.store.view("Text/W_embedding")
        self.logits = None

    @property
    def concat_features(self):
        """tf.concat([images_features, texts_features], 1) (:95).  The training path never builds it
        (the dense layer reads both halves in place); it is materialised on demand for callers."""
        h = self.net.head
        out = torch.empty(h.B, h.im + h.tx, device=self.net.device)
        ops.copy2d(h.im_feat, h.im_feat.stride(0), out, h.im + h.tx, h.B, h.im)
        ops.copy2d(h.tx_feat, h.tx_feat.stride(0), out[:, h.im:], h.im + h.tx, h.B, h.tx)
        return out


def train_deep_sentiment(checkpoints_dir, train_dir, num_steps, *, config=None, quiet=False):
    """Fine tune the inception model, retraining the last layer (im_text_rnn_model.py:107-169)."""
    model = DeepSentiment(dict(_CONFIG, **(config or {})))
    init_fn = get_init_fn(checkpoints_dir)
    if init_fn is not None:
        init_fn(model.net)
    return run_training(model, train_dir, num_steps, quiet=quiet)


def evaluate_deep_sentiment(checkpoint_dir, log_dir, mode, num_evals, *, config=None, quiet=False):
    """Accuracy of the newest checkpoint (im_text_rnn_model.py:171-207)."""
    from ..training import run_evaluation
    model = DeepSentiment(dict(_CONFIG, mode=mode, **(config or {})))
    return run_evaluation(model, checkpoint_dir, log_dir, mode, num_evals, quiet=quiet)
