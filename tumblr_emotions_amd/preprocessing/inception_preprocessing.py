"""`preprocess_for_eval` of slim/preprocessing/inception_preprocessing.py:237-275 in NumPy -- the ONLY
image pipeline the reference's training path uses (load_batch_with_text is always called with
is_training=False, image_model/im_model.py:78,102): uint8 -> [0,1] float, central crop 87.5 %, bilinear
resize (align_corners=False, TF-1.x legacy sampling src = dst * in/out), then (x - 0.5) * 2.

[TF-sem] tf.image.central_crop: start = int((size - size*fraction) / 2), extent = size - 2*start.
Parity unpinned: no TensorFlow output is available to compare the resize against."""
import numpy as np


def central_crop(image, central_fraction):
    h, w = image.shape[0], image.shape[1]
    h0 = int((h - h * central_fraction) / 2)
    w0 = int((w - w * central_fraction) / 2)
    return image[h0:h0 + (h - 2 * h0), w0:w0 + (w - 2 * w0)]


def resize_bilinear(image, height, width):
    """[H,W,C] float32 -> [height,width,C]; tf.image.resize_bilinear(align_corners=False), TF 1.x."""
    h, w = image.shape[0], image.shape[1]

    def axis(n_in, n_out):
        src = np.arange(n_out, dtype=np.float32) * np.float32(n_in / n_out)
        lo = np.floor(src).astype(np.int64)
        hi = np.minimum(lo + 1, n_in - 1)
        return lo, hi, (src - lo).astype(np.float32)

    y0, y1, fy = axis(h, height)
    x0, x1, fx = axis(w, width)
    top = image[y0][:, x0] + (image[y0][:, x1] - image[y0][:, x0]) * fx[None, :, None]
    bot = image[y1][:, x0] + (image[y1][:, x1] - image[y1][:, x0]) * fx[None, :, None]
    return (top + (bot - top) * fy[:, None, None]).astype(np.float32)


def preprocess_for_eval(image, height, width, central_fraction=0.875):
    if image.dtype != np.float32:
        image = image.astype(np.float32) / np.float32(np.iinfo(image.dtype).max)     # convert_image_dtype
    if central_fraction:
        image = central_crop(image, central_fraction)
    if height and width:
        image = resize_bilinear(image, height, width)
    return (image - np.float32(0.5)) * np.float32(2.0)


def preprocess_image(image, height, width, is_training=False):
    if is_training:
        raise NotImplementedError("the reference's training path never uses the train-time augmentation")
    return preprocess_for_eval(image, height, width)
