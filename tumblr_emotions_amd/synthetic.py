"""Synthetic (image, token-id, seq_len, label) batches with the shapes/ranges of the reference's
input pipeline.  There is no network and no Tumblr data here, so this stands in
for `get_split_with_text` + `load_batch_with_text` (datasets/convert_to_dataset.py:117,
image_model/im_model.py:78-116):
  images  [B,224,224,3] fp32 NHWC, U(-1,1)      = range of preprocess_for_eval
                                                  (slim/preprocessing/inception_preprocessing.py:273-274)
  texts   [B,T] int64 in [0,V), positions >= seq_len hold the pad/unk id V
                                                  (text_model/text_preprocessing.py:98,104)
  seq_lens[B] int64 in [6,T]                      (> 5 in-vocabulary words, text_preprocessing.py:11,82)
  labels  [B] int64 in [0, nb_emotions)
"""
import numpy as np
import torch


class SyntheticDataset:
    """Shape of slim's Dataset object as far as the trainers read it (.num_samples, .num_classes)."""

    def __init__(self, num_samples=50000, num_classes=15):
        self.num_samples = num_samples
        self.num_classes = num_classes


def synthetic_batch_numpy(batch, post_size, vocab, nb_emotions=15, image_size=224, seed=0, with_images=True):
    rng = np.random.RandomState(seed)
    out = {}
    if with_images:
        out["images"] = rng.uniform(-1, 1, size=(batch, image_size, image_size, 3)).astype(np.float32)
    seq_len = rng.randint(min(6, post_size), post_size + 1, size=batch).astype(np.int64)
    ids = rng.randint(0, vocab, size=(batch, post_size)).astype(np.int64)
    ids[np.arange(post_size)[None, :] >= seq_len[:, None]] = vocab
    out["texts"], out["seq_lens"] = ids, seq_len
    out["labels"] = rng.randint(0, nb_emotions, size=batch).astype(np.int64)
    out["post_ids"] = np.arange(batch, dtype=np.int64)
    out["days"] = rng.randint(0, 7, size=batch).astype(np.int64)
    return out


def to_device(batch, device="cuda", rank=0, world=1):
    """Move a numpy batch to the device; under data parallelism rank r takes the r-th contiguous slice
    of the SAME seeded global batch."""
    out = {}
    for k, v in batch.items():
        n = v.shape[0]
        lo, hi = rank * n // world, (rank + 1) * n // world
        out[k] = torch.from_numpy(np.ascontiguousarray(v[lo:hi])).to(device)
    return out
