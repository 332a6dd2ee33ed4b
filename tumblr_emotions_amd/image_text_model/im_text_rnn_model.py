"""Deep Sentiment (joint image + text) model and trainer with the reference's signatures
(image_text_model/im_text_rnn_model.py:38-169)."""
import numpy as np
import torch

from .. import ops
from ..image_model.im_model import get_init_fn
from ..net import SentimentNet
from ..text_model.text_preprocessing import resolve_embedding
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
        post = config.get('post_size', _POST_SIZE)
        # GloVe file -> [V, D] + zero <ukn> row; V and D come from the file (:71-76)
        embedding, vocab, dim, self.word_to_id = resolve_embedding(config, embedding)
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


# ---- inference-only analyses of the trained joint model (im_text_rnn_model.py:342-575) ----------
# Each one restores the newest checkpoint of `checkpoint_dir` into a validation-mode model (BatchNorm on moving
# statistics, no dropout), runs forward passes on the HIP kernels and writes the same .npy files under `out_dir`
# ('data' in the reference).  `config` overrides _CONFIG as everywhere else in this module.

def _restored_validation_model(checkpoint_dir, config):
    from ..training import latest_checkpoint, load_checkpoint
    model = DeepSentiment(dict(_CONFIG, **dict(config or {}, mode='validation')))
    path = latest_checkpoint(checkpoint_dir)
    if path is None:
        raise FileNotFoundError("no checkpoint in %s" % checkpoint_dir)
    load_checkpoint(model, path)
    return model


def _forward_batches(model, nb_batches, want_features=False):
    """Yield (logits, labels, days, post_ids[, concat_features]) as numpy arrays for `nb_batches` batches."""
    for i in range(nb_batches):
        batch = model.next_batch(10 ** 6 + i)
        logits = model.net.predict(batch, is_training=False)
        out = [logits.cpu().numpy(), batch["labels"].cpu().numpy(), model.days.cpu().numpy(),
               model.post_ids.cpu().numpy()]
        if want_features:
            out.append(model.concat_features.cpu().numpy())
        yield out


def _save(out_dir, **arrays):
    import os
    os.makedirs(out_dir, exist_ok=True)
    for name, a in arrays.items():
        np.save(os.path.join(out_dir, name + ".npy"), a)


def correlation_matrix(nb_batches, checkpoint_dir, *, config=None, out_dir='data'):
    """Logits and labels of `nb_batches` validation batches -> posts_logits.npy, posts_labels.npy (:342-376)."""
    model = _restored_validation_model(checkpoint_dir, config)
    rows = list(_forward_batches(model, nb_batches))
    posts_logits = np.vstack([r[0] for r in rows])
    posts_labels = np.hstack([r[1] for r in rows])
    _save(out_dir, posts_logits=posts_logits, posts_labels=posts_labels)
    return posts_logits, posts_labels


def day_of_week_trend(checkpoint_dir, *, config=None, out_dir='data'):
    """Logits, labels, day-of-week and post id of every whole validation batch -> posts_*_week.npy (:531-575)."""
    model = _restored_validation_model(checkpoint_dir, config)
    nb_batches = model.dataset.num_samples // model.config['batch_size']
    rows = list(_forward_batches(model, nb_batches))
    posts_logits = np.vstack([r[0] for r in rows])
    posts_labels, posts_days, posts_ids = (np.hstack([r[k] for r in rows]) for k in (1, 2, 3))
    _save(out_dir, posts_logits_week=posts_logits, posts_labels_week=posts_labels, posts_days_week=posts_days,
          posts_ids_week=posts_ids)
    return posts_logits, posts_labels, posts_days, posts_ids


def outliers_detection(checkpoint_dir, *, config=None, out_dir='data'):
    """Posts whose concatenated image+text feature vector lies farthest (Euclidean) from the validation mean
    (:478-529).  As in the reference the maximum is tracked per batch SLOT k (not globally): slot k keeps the
    largest distance seen at position k of any batch, with that post's id and logits."""
    model = _restored_validation_model(checkpoint_dir, config)
    batch_size = model.config['batch_size']
    nb_batches = model.dataset.num_samples // batch_size
    dense_mean = None
    for i, row in enumerate(_forward_batches(model, nb_batches, want_features=True)):
        m = row[4].mean(axis=0, dtype=np.float64)
        dense_mean = m if dense_mean is None else (i * dense_mean + m) / (i + 1)       # running mean of batch means
    max_norms = np.zeros(batch_size)
    max_post_ids = np.zeros(batch_size)
    max_logits = np.zeros((batch_size, model.dataset.num_classes))
    model._records = None                    # second pass over the same validation stream
    for logits, _, _, post_ids, feats in _forward_batches(model, nb_batches, want_features=True):
        dist = np.linalg.norm(feats - dense_mean, axis=1)
        better = dist > max_norms
        max_norms[better], max_post_ids[better], max_logits[better] = dist[better], post_ids[better], logits[better]
    _save(out_dir, max_norms=max_norms, max_post_ids=max_post_ids, max_logits=max_logits)
    return max_norms, max_post_ids, max_logits


def word_most_relevant(top_words, num_classes, checkpoint_dir, *, config=None, out_dir='data', vocabulary=None):
    """Score single words: a post made of word w alone (sequence length 1, the rest padding) next to an all-zero
    image, through the trained joint model; scores[i] = logits for top_words[i] (:378-475; the reference
    reads an undefined `fc_size` at :436 -- here it comes from the config like everywhere else).
    Returns (scores, vocabulary, word_to_id); `vocabulary` defaults to the GloVe file of the config."""
    model = _restored_validation_model(checkpoint_dir, config)
    assert model.dataset.num_classes == num_classes or num_classes is None
    cfg = model.config
    if vocabulary is None:
        if model.word_to_id is not None:      # the GloVe vocabulary the constructor loaded
            vocabulary = [w for w, _ in sorted(model.word_to_id.items(), key=lambda kv: kv[1]) if w != '<ukn>']
        else:                                 # synthetic run: ids stand for themselves
            vocabulary = [str(i) for i in range(model.net.text.V - 1)]
    word_to_id = dict(zip(vocabulary, range(len(vocabulary))))
    word_to_id['<ukn>'] = len(vocabulary)
    pad = model.net.text.V - 1
    post, batch_size = model.net.text.T, 50                       # the reference hard-codes 50 (:398)
    top_words = np.asarray(top_words, dtype=np.int64)
    scores = []
    dev = model.net.device
    for i in range(len(top_words) // batch_size):                 # a ragged tail is dropped, as in the reference
        texts = np.full((batch_size, post), pad, dtype=np.int64)
        texts[:, 0] = top_words[i * batch_size:(i + 1) * batch_size]
        batch = {"images": torch.zeros(batch_size, 224, 224, 3, device=dev),
                 "texts": torch.from_numpy(texts).to(dev),
                 "seq_lens": torch.ones(batch_size, dtype=torch.int64, device=dev),
                 "labels": torch.zeros(batch_size, dtype=torch.int64, device=dev)}
        scores.append(model.net.predict(batch, is_training=False).cpu().numpy())
    scores = np.vstack(scores) if scores else np.zeros((0, model.dataset.num_classes), np.float32)
    _save(out_dir, top_words_scores=scores, top_words=top_words)
    return scores, vocabulary, word_to_id
