"""Text-only model + trainer with the reference's signatures (text_model/text_embedding.py:37-150)."""
import numpy as np
import torch

from ..net import SentimentNet
from .text_preprocessing import resolve_embedding
from ..training import SyntheticInput, run_training

_RANDOM_SEED = 0
_CONFIG = {'mode': 'train',
           'dataset_dir': 'data',
           'text_dir': 'text_model',
           'emb_dir': 'embedding_weights',
           'filename': 'glove.6B.50d.txt',
           'initial_lr': 1e-3,
           'decay_factor': 0.3,
           'batch_size': 64,
           'rnn_size': 1024}                    # keys verbatim from text_embedding.py:16-24
_POST_SIZE = 50                                 # datasets/convert_to_dataset.py text[50]


class TextModel(SyntheticInput):
    def __init__(self, config, nb_emotions=15, embedding=None, device="cuda", **net_kw):
        self.config = config
        self.learning_rate = config['initial_lr']
        post = config.get('post_size', _POST_SIZE)
        # GloVe file -> [V, D] + zero <ukn> row; V and D come from the file (:61-66)
        embedding, vocab, dim, self.word_to_id = resolve_embedding(config, embedding)
        self._init_input(config, post, vocab, nb_emotions, False, device)
        self.nb_emotions = self.dataset.num_classes
        for key in ("train_all", "trainable_embedding"):      # optional fine-tuning switches (not in the reference _CONFIG)
            if key in config:
                net_kw.setdefault(key, bool(config[key]))
        self.net = SentimentNet(mode="text", nb_emotions=self.nb_emotions, rnn_size=config['rnn_size'],
                                vocab_size=vocab, embedding_dim=dim, post_size=post, device=device, **net_kw)
        self.net.initialize(seed=config.get('seed', 1))
        if embedding is not None:               # the step-0 embedding_init assign (:131-132)
            self.net.store.view("Text/W_embedding").copy_(torch.from_numpy(embedding))
        self.embedding = self.net.store.view("Text/W_embedding")
        self.logits = None


def train_text_model(train_dir, num_steps, *, config=None, quiet=False):
    """Train rnn text model (text_embedding.py:89-150)."""
    model = TextModel(dict(_CONFIG, **(config or {})))
    return run_training(model, train_dir, num_steps, quiet=quiet)


def evaluate_text_model(checkpoint_dir, log_dir, mode, num_evals, *, config=None, quiet=False):
    """Accuracy of the newest checkpoint (text_embedding.py:152-187)."""
    from ..training import run_evaluation
    model = TextModel(dict(_CONFIG, mode=mode, **(config or {})))
    return run_evaluation(model, checkpoint_dir, log_dir, mode, num_evals, quiet=quiet)
