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


def get_init_fn(checkpoints_dir, model_name='inception_v1.npz'):
    """Warm start (im_model.py:118-137): restore every slim model variable except those under
    InceptionV1/Logits / InceptionV1/AuxLogits.  The checkpoint is a .npz of TF-named arrays (a
    reader for TF's own checkpoint format is out of scope).  Returns fn(net) or None."""
    path = os.path.join(checkpoints_dir or "", model_name)
    if not checkpoints_dir or not os.path.exists(path):
        return None
    exclusions = ("InceptionV1/Logits", "InceptionV1/AuxLogits")

    def init_fn(net):
        sd = {k: v for k, v in np.load(path).items() if not k.startswith(exclusions)}
        net.load_state_dict(sd, strict=False)
    return init_fn


class ImageModel(SyntheticInput):
    def __init__(self, config, nb_emotions=15, device="cuda", **net_kw):
        self.config = config
        if config.get('final_endpoint', 'Mixed_5c') != 'Mixed_5c':
            raise NotImplementedError("final_endpoint must be Mixed_5c")
        self.learning_rate = config['initial_lr']
        self._init_input(config, config.get('post_size', 50), config.get('vocab_size', 400000), nb_emotions, True, device)
        self.nb_emotions = self.dataset.num_classes
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
