"""Writers/readers of the dataset side files, with the reference's names
(datasets/dataset_utils.py:28-76,99-151)."""
import os

from .tfrecord import encode_example

LABELS_FILENAME = 'labels.txt'


def image_to_tfexample_with_text(image_data, image_format, height, width, text_data, seq_len, class_id, post_id, day):
    """Serialized tf.train.Example with the reference's feature names (dataset_utils.py:65-76)."""
    return encode_example({
        'image/encoded': image_data,
        'image/format': image_format,
        'image/class/label': class_id,
        'image/height': height,
        'image/width': width,
        'text': list(text_data),
        'seq_len': seq_len,
        'post_id': post_id,
        'day': day,
    })


def write_label_file(labels_to_class_names, dataset_dir, photos_subdir, filename=LABELS_FILENAME):
    with open(os.path.join(dataset_dir, photos_subdir, filename), 'w') as f:
        for label in labels_to_class_names:
            f.write('%d:%s\n' % (label, labels_to_class_names[label]))


def has_labels(dataset_dir, photos_subdir, filename=LABELS_FILENAME):
    return os.path.exists(os.path.join(dataset_dir, photos_subdir, filename))


def read_label_file(dataset_dir, photos_subdir, filename=LABELS_FILENAME):
    with open(os.path.join(dataset_dir, photos_subdir, filename), 'rb') as f:
        lines = [l for l in f.read().decode().split('\n') if l]
    out = {}
    for line in lines:
        i = line.index(':')
        out[int(line[:i])] = line[i + 1:]
    return out
