"""`get_split_with_text` without TensorFlow (datasets/convert_to_dataset.py:117-198): describes a split
of the Tumblr dataset (sharded TFRecords of image + token ids) and iterates its decoded examples."""
import glob
import io
import os

import numpy as np

from . import dataset_utils
from .tfrecord import decode_example, read_records

_FILE_PATTERN = 'tumblr_%s_*.tfrecord'                  # convert_to_dataset.py:29
_TRAIN_VALID_FILENAME = 'train_valid_split.txt'         # :32
_POST_SIZE = 50                                         # :34
_NUM_SHARDS = 5                                         # datasets/convert_images_tfrecords.py


def dataset_filename(dataset_dir, subdir, split_name, shard_id, num_shards=_NUM_SHARDS):
    """Shard naming of convert_images_tfrecords.py:110-113."""
    return os.path.join(dataset_dir, subdir, 'tumblr_%s_%05d-of-%05d.tfrecord' % (split_name, shard_id, num_shards))


class Dataset:
    """The fields of slim.dataset.Dataset the trainers read, plus an example iterator."""

    def __init__(self, data_sources, num_samples, num_classes, labels_to_names):
        self.data_sources, self.num_samples = data_sources, num_samples
        self.num_classes, self.labels_to_names = num_classes, labels_to_names

    def examples(self, verify_crc=False, decode_image=None):
        """Yields dicts: image (uint8 HxWx3), text (int64[50]), seq_len, label, post_id, day
        (the items_to_handlers of convert_to_dataset.py:163-170).  decode_image(index) -> bool lets a
        data-parallel reader skip the JPEG decode of records that belong to other ranks (image = None)."""
        from PIL import Image
        idx = -1
        for path in self.data_sources:
            for rec in read_records(path, verify=verify_crc):
                idx += 1
                ex = decode_example(rec)
                img = None
                if decode_image is None or decode_image(idx):
                    img = np.asarray(Image.open(io.BytesIO(ex['image/encoded'][0])).convert('RGB'))
                text = np.zeros(_POST_SIZE, np.int64)
                t = ex.get('text', [])
                text[:len(t)] = t
                yield dict(image=img, text=text, seq_len=int(ex.get('seq_len', [0])[0]),
                           label=int(ex.get('image/class/label', [0])[0]), post_id=int(ex.get('post_id', [0])[0]),
                           day=int(ex.get('day', [0])[0]))


def get_split_with_text(split_name, dataset_dir, photos_subdir='photos', tfrecords_subdir='tfrecords',
                        file_pattern=None, reader=None):
    pattern = os.path.join(dataset_dir, tfrecords_subdir, (file_pattern or _FILE_PATTERN) % split_name)
    labels_to_names = None
    if dataset_utils.has_labels(dataset_dir, photos_subdir):
        labels_to_names = dataset_utils.read_label_file(dataset_dir, photos_subdir)
    with open(os.path.join(dataset_dir, photos_subdir, _TRAIN_VALID_FILENAME), 'rb') as f:
        lines = [l for l in f.read().decode().split('\n') if l]
    sizes = {}
    for line in lines:
        i = line.index(':')
        sizes[line[:i]] = int(line[i + 1:])
    return Dataset(sorted(glob.glob(pattern)), sizes[split_name], len(labels_to_names), labels_to_names)
