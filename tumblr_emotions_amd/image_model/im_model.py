"""Image-only model + trainer with the reference's signatures (image_model/im_model.py:139-225)."""
import os

import numpy as np

from ..net import SentimentNet
from ..training import SyntheticInput, run_training

_RANDOM_SEED = 0
_CONFIG = {'mode': 'train',
           'dataset_dir': 'data',
           'initial_lr': 1e-3,
           'decay_factor': 0.3,
           'batch_size': 64,
           'final_endpoint': 'Mixed_5c'}        # keys verbatim from im_model.py:20-25


def download_pretrained_model(url, checkpoint_dir):
    """im_model.py:27-36.  This build has no network access; place inception_v1.ckpt-derived
    arrays in `checkpoint_dir/inception_v1.npz` instead (see get_init_fn)."""
    raise RuntimeError("no network: cannot download %s into %s" % (url, checkpoint_dir))


def get_init_fn(checkpoints_dir, model_name=None):
    """Warm start (im_model.py:118-137): restore every slim model variable except those under
    InceptionV1/Logits / InceptionV1/AuxLogits.  Looks for `inception_v1.ckpt` (TensorFlow's V1 checkpoint,
    read by checkpoint_tf.read_tf_v1_checkpoint without TensorFlow) and then for `inception_v1.npz` (a dump
    of TF-named arrays) in `checkpoints_dir`.  Returns fn(net) or None when neither exists."""
    exclusions = ("InceptionV1/Logits", "InceptionV1/AuxLogits")
    candidates = [model_name] if model_name else ["inception_v1.ckpt", "inception_v1.npz"]
    path = next((os.path.join(checkpoints_dir, n) for n in candidates
                 if checkpoints_dir and os.path.exists(os.path.join(checkpoints_dir, n))), None)
    if path is None:
        return None

    def init_fn(net):
        if path.endswith(".npz"):
            sd = {k: v for k, v in np.load(path).items() if not k.startswith(exclusions)}
        else:
            from ..checkpoint_tf import read_tf_v1_checkpoint
            sd = read_tf_v1_checkpoint(path, names=lambda n: n.startswith("InceptionV1/") and not n.startswith(exclusions))
        wanted = [e for e in net.store.tf_names() if e.startswith("InceptionV1/") and not e.startswith(exclusions)]
        seen = set(net.load_state_dict(sd, strict=False))
        missing = [n for n in wanted if n not in seen]
        if missing:      # a silent random-init "fine-tune" is worse than stopping (im_model.py:118-137 restores them all)
            raise KeyError("warm start from %s: %d of %d InceptionV1 variables not found, e.g. %s"
                           % (path, len(missing), len(wanted), missing[:3]))
        print("Restored %d variables from %s" % (len(seen), path))
    return init_fn


class ImageModel(SyntheticInput):
    def __init__(self, config, nb_emotions=15, device="cuda", **net_kw):
        self.config = config
        if config.get('final_endpoint', 'Mixed_5c') != 'Mixed_5c':
            raise NotImplementedError("final_endpoint must be Mixed_5c")
        self.learning_rate = config['initial_lr']
        self._init_input(config, config.get('post_size', 50), config.get('vocab_size', 400000), nb_emotions, True, device)
        self.nb_emotions = self.dataset.num_classes
        for key in ("train_all", "trainable_embedding"):      # optional fine-tuning switches (not in the reference _CONFIG)
            if key in config:
                net_kw.setdefault(key, bool(config[key]))
        self.net = SentimentNet(mode="image", nb_emotions=self.nb_emotions, device=device, **net_kw)
        self.net.initialize(seed=config.get('seed', 1))
        self.logits = None


def train_image_model(checkpoints_dir, train_dir, num_steps, *, config=None, quiet=False):
    """Fine tune the Image model, retraining Mixed_5c (im_model.py:166-225)."""
    model = ImageModel(dict(_CONFIG, **(config or {})))
    init_fn = get_init_fn(checkpoints_dir)
    if init_fn is not None:
        init_fn(model.net)
    return run_training(model, train_dir, num_steps, quiet=quiet)


def evaluate_image_model(checkpoint_dir, log_dir, mode, num_evals, *, config=None, quiet=False):
    """Accuracy of the newest checkpoint on `mode` ('train' | 'validation') over num_evals batches
    (im_model.py:227-262)."""
    from ..training import run_evaluation
    model = ImageModel(dict(_CONFIG, mode=mode, **(config or {})))
    return run_evaluation(model, checkpoint_dir, log_dir, mode, num_evals, quiet=quiet)


def load_batch_with_text(dataset, batch_size=32, shuffle=True, height=299, width=299, is_training=False,
                         device="cuda", rank=0, world=1, seed=0, loop=True, max_token_id=None, num_classes=None):
    """Generator of training batches from a `datasets.convert_to_dataset.Dataset` -- the role of
    load_batch_with_text + tf.train.batch in the reference (im_model.py:78-116): decode the JPEG, apply the
    EVAL preprocessing (is_training=False is what every reference call site uses, :78,102), batch.
    Yields dicts of device tensors: images [B,height,width,3] f32 in [-1,1], texts [B,50] i64, seq_lens,
    labels, post_ids, days.  Under data parallelism every rank reads the same stream and keeps examples
    rank, rank+world, ... (disjoint shards of one global order; the others are skipped BEFORE the JPEG is
    decoded).  max_token_id / num_classes: a record whose token ids exceed the embedding table (ids index the
    vocabulary the dataset was converted with; id == vocabulary size is '<ukn>') or whose label is out of
    range raises -- the gather would otherwise read zero rows / the loss kernel out of bounds."""
    import torch
    from ..preprocessing.inception_preprocessing import preprocess_image
    rng = np.random.RandomState(seed)
    buf = {k: [] for k in ("images", "texts", "seq_lens", "labels", "post_ids", "days")}
    while True:
        sources = list(dataset.data_sources)
        if shuffle:
            rng.shuffle(sources)
        view = type(dataset)(sources, dataset.num_samples, dataset.num_classes, dataset.labels_to_names)
        n = 0
        for i, ex in enumerate(view.examples(decode_image=lambda idx: idx % world == rank)):
            if i % world != rank:
                continue
            n += 1
            if max_token_id is not None and int(np.max(ex["text"])) > max_token_id:
                raise ValueError("token id %d in the dataset exceeds the embedding table (%d rows + <ukn>): the "
                                 "dataset was converted with a different vocabulary" % (int(np.max(ex["text"])), max_token_id))
            if num_classes is not None and not 0 <= int(ex["label"]) < num_classes:
                raise ValueError("label %d outside [0, %d)" % (int(ex["label"]), num_classes))
            buf["images"].append(preprocess_image(ex["image"], height, width, is_training=is_training))
            buf["texts"].append(ex["text"])
            for k, s in (("seq_lens", "seq_len"), ("labels", "label"), ("post_ids", "post_id"), ("days", "day")):
                buf[k].append(ex[s])
            if len(buf["labels"]) == batch_size:
                order = rng.permutation(batch_size) if shuffle else np.arange(batch_size)
                out = {"images": torch.from_numpy(np.stack(buf["images"])[order]).to(device),
                       "texts": torch.from_numpy(np.stack(buf["texts"])[order]).to(device)}
                for k in ("seq_lens", "labels", "post_ids", "days"):
                    out[k] = torch.from_numpy(np.asarray(buf[k], np.int64)[order]).to(device)
                buf = {k: [] for k in buf}
                yield out
        if not loop or n == 0:
            return
