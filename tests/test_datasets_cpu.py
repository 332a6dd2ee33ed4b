"""TFRecord / tf.Example codec, dataset side files and the eval preprocessing (row 8f-2), CPU only.
Pinned: CRC-32C check value; bilinear resize against scipy/torch on the align_corners=False legacy
grid; everything else by round trips (no TensorFlow-written artefact exists in the reference tree)."""
import io
import os

import numpy as np
import torch
import torch.nn.functional as F

from tumblr_emotions_amd.datasets import convert_to_dataset as cd
from tumblr_emotions_amd.datasets import dataset_utils as du
from tumblr_emotions_amd.datasets import tfrecord as T
from tumblr_emotions_amd.preprocessing import inception_preprocessing as ip


def test_crc32c_check_value_and_mask():
    assert T.crc32c(b"123456789") == 0xE3069283          # CRC-32C (Castagnoli) standard check value
    assert T.crc32c(b"") == 0
    assert T.masked_crc(b"") == 0xA282EAD8               # mask of crc 0


def test_example_codec_round_trip_and_negative_int64():
    feats = {'image/encoded': b'\xff\xd8abc', 'image/format': b'jpg', 'image/class/label': 7,
             'text': list(range(45)) + [400000] * 5, 'seq_len': 45, 'post_id': -3, 'day': 6, 'f': [0.25, -1.5]}
    got = T.decode_example(T.encode_example(feats))
    assert got['image/encoded'] == [b'\xff\xd8abc'] and got['text'] == feats['text']
    assert got['post_id'] == [-3] and got['f'] == [0.25, -1.5] and got['image/class/label'] == [7]


def _make_dataset(root, n_train=7, n_valid=3):
    from PIL import Image
    os.makedirs(os.path.join(root, "photos"))
    os.makedirs(os.path.join(root, "tfrecords"))
    du.write_label_file({0: "happy", 1: "sad", 2: "angry"}, root, "photos")
    with open(os.path.join(root, "photos", cd._TRAIN_VALID_FILENAME), "w") as f:
        f.write("train:%d\nvalidation:%d\n" % (n_train, n_valid))
    rng = np.random.RandomState(0)
    truth = {}
    for split, n in (("train", n_train), ("validation", n_valid)):
        recs = [[], []]
        for i in range(n):
            img = rng.randint(0, 256, size=(40 + i, 52, 3)).astype(np.uint8)
            b = io.BytesIO()
            Image.fromarray(img).save(b, format="PNG")        # lossless, so pixels can be compared exactly
            text = rng.randint(0, 100, size=50).tolist()
            recs[i % 2].append(du.image_to_tfexample_with_text(b.getvalue(), b'png', img.shape[0], img.shape[1], text,
                                                               10 + i, i % 3, 1000 + i, i % 7))
            truth[(split, 1000 + i)] = (img, text, 10 + i, i % 3, i % 7)
        for shard in range(2):
            T.write_records(cd.dataset_filename(root, "tfrecords", split, shard, 2), recs[shard])
    return truth


def test_dataset_write_read_round_trip(tmp_path):
    truth = _make_dataset(str(tmp_path))
    ds = cd.get_split_with_text("train", str(tmp_path))
    assert ds.num_samples == 7 and ds.num_classes == 3 and ds.labels_to_names[1] == "sad"
    assert [os.path.basename(p) for p in ds.data_sources] == ["tumblr_train_00000-of-00002.tfrecord",
                                                              "tumblr_train_00001-of-00002.tfrecord"]
    seen = 0
    for ex in ds.examples(verify_crc=True):
        img, text, seq_len, label, day = truth[("train", ex["post_id"])]
        assert np.array_equal(ex["image"], img) and ex["text"].tolist() == text
        assert (ex["seq_len"], ex["label"], ex["day"]) == (seq_len, label, day)
        seen += 1
    assert seen == 7
    assert cd.get_split_with_text("validation", str(tmp_path)).num_samples == 3


def test_corrupt_record_is_detected(tmp_path):
    p = str(tmp_path / "x.tfrecord")
    T.write_records(p, [b"hello world"])
    raw = bytearray(open(p, "rb").read())
    raw[14] ^= 1
    open(p, "wb").write(bytes(raw))
    try:
        list(T.read_records(p, verify=True))
        assert False, "corruption not detected"
    except IOError:
        pass


def test_preprocess_for_eval_geometry_and_range():
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, size=(300, 200, 3)).astype(np.uint8)
    out = ip.preprocess_for_eval(img, 224, 224)
    assert out.shape == (224, 224, 3) and out.dtype == np.float32
    assert out.min() >= -1.0 and out.max() <= 1.0
    crop = ip.central_crop(img, 0.875)
    assert crop.shape == (300 - 2 * 18, 200 - 2 * 12, 3)          # start = int((300-262.5)/2) = 18, int(12.5) = 12
    # identity resize and a pure 2x down-sample pick source pixels exactly (src = dst * in/out)
    x = rng.uniform(size=(8, 6, 2)).astype(np.float32)
    np.testing.assert_array_equal(ip.resize_bilinear(x, 8, 6), x)
    np.testing.assert_allclose(ip.resize_bilinear(x, 4, 3), x[::2, ::2], rtol=0, atol=0)
    # up-sampling agrees with an independent implementation of the same legacy grid
    up = ip.resize_bilinear(x, 16, 9)
    ys = np.arange(16) * (8 / 16)
    xs = np.arange(9) * (6 / 9)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    y1, x1 = np.minimum(y0 + 1, 7), np.minimum(x0 + 1, 5)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    ref = (x[y0][:, x0] * (1 - fy) * (1 - fx) + x[y0][:, x1] * (1 - fy) * fx + x[y1][:, x0] * fy * (1 - fx)
           + x[y1][:, x1] * fy * fx)
    np.testing.assert_allclose(up, ref, atol=1e-6)


def test_crc32c_rfc3720_vectors_and_mask():
    """CRC-32C against the iSCSI test vectors (RFC 3720 appendix B.4), for the table-driven implementation of the
    package AND the bit-wise one the fixture generator uses; the TFRecord mask is rot-right-15 + 0xa282ead8."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_handmade_fixtures import crc32c_bitwise, masked
    vectors = [(bytes(32), 0x8A9136AA), (b"\xff" * 32, 0x62A8AB43), (bytes(range(32)), 0x46DD794E),
               (bytes(range(31, -1, -1)), 0x113FDB5C), (b"123456789", 0xE3069283)]
    for data, want in vectors:
        assert T.crc32c(data) == want and crc32c_bitwise(data) == want
    c = T.crc32c(b"foo")
    assert T.masked_crc(b"foo") == masked(b"foo") == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def test_handmade_tf_example_records_decode():
    """tests/golden/handmade_examples.tfrecord: two framed tf.train.Example records assembled byte by byte by
    tests/golden/make_handmade_fixtures.py (independent of this package) in encodings our writer never produces."""
    path = os.path.join(os.path.dirname(__file__), "golden", "handmade_examples.tfrecord")
    recs = list(T.read_records(path, verify=True))
    assert len(recs) == 2 and recs[1] == b""
    ex = T.decode_example(recs[0])
    assert ex["text"] == [5, 300, 70000, -1]                       # unpacked Int64List, 10-byte negative varint
    assert ex["image/encoded"] == [b"\x89PNG-not-really", b"second"]
    assert ex["image/format"] == [b"png"]                          # value before key, unknown field skipped
    assert ex["seq_len"] == [300, 2]
    assert ex["weights"] == [0.5, -2.25]                           # unpacked FloatList
    assert ex["k" * 200] == [1]
    assert len(ex) == 6
    assert T.decode_example(recs[1]) == {}
    # our own writer's encoding of the same content decodes to the same dict (packed, sorted)
    again = T.decode_example(T.encode_example({k: v for k, v in ex.items()}))
    assert again == ex


def test_protobuf_runtime_serialised_examples_decode():
    """tests/golden/protobuf_examples.tfrecord: tf.train.Example messages of the reference's dataset schema
    (/root/reference/datasets/convert_to_dataset.py:148-161, dataset_utils.py:65-76) serialised by the OFFICIAL protobuf
    runtime from descriptors declared in tests/golden/make_protobuf_fixtures.py -- an encoder this repository did not write.
    The reader must return exactly the values that went into the runtime (protobuf_fixtures.json): 50-id texts with 2- and
    3-byte varints, a negative int64, empty bytes, floats incl. a denormal, an empty list, a feature with no kind, and a
    record from a "newer writer" whose extra Feature / Example fields must be skipped."""
    import json
    here = os.path.join(os.path.dirname(__file__), "golden")
    fx = json.load(open(os.path.join(here, "protobuf_fixtures.json")))
    recs = list(T.read_records(os.path.join(here, "protobuf_examples.tfrecord"), verify=True))
    assert len(recs) == len(fx["examples"]) == 4
    for rec, want in zip(recs, fx["examples"]):
        got = T.decode_example(rec)
        assert sorted(got) == sorted(want)
        for k, v in want.items():
            if v and isinstance(v[0], str):
                assert [x.hex() for x in got[k]] == v, k
            elif v and isinstance(v[0], float):
                assert [np.float32(x) for x in got[k]] == [np.float32(x) for x in v], k
            else:
                assert got[k] == v, k
    # the dataset record round-trips through this package's own writer to the same content
    first = T.decode_example(recs[0])
    assert T.decode_example(T.encode_example(first)) == first
    assert len(first["text"]) == 50 and first["seq_len"] == [37] and first["image/format"] == [b"jpg"]
