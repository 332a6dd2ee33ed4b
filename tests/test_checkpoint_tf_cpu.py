"""TensorFlow V1 checkpoint reader (row 8f-4): table framing, prefix-compressed keys, several data blocks,
sliced tensors, checksums, snappy blocks.  Pinned against this build's own writer and hand-assembled bytes only:
no TensorFlow-written file exists here (see the module docstring)."""
import os
import struct

import numpy as np
import pytest

from tumblr_emotions_amd import checkpoint_tf as C
from tumblr_emotions_amd.datasets.tfrecord import masked_crc


def _tensors(rng):
    return {
        "InceptionV1/Conv2d_1a_7x7/weights": rng.normal(size=(7, 7, 3, 64)).astype(np.float32),
        "InceptionV1/Conv2d_1a_7x7/BatchNorm/beta": rng.normal(size=64).astype(np.float32),
        "InceptionV1/Mixed_3b/Branch_1/Conv2d_0a_1x1/weights": rng.normal(size=(1, 1, 192, 96)).astype(np.float32),
        "InceptionV1/Logits/Conv2d_0c_1x1/biases": rng.normal(size=1001).astype(np.float32),
        "global_step": np.array(123456789012, dtype=np.int64),
        "some/int32": np.arange(-5, 6, dtype=np.int32),
        "some/double": rng.normal(size=(3, 2)),
    }


@pytest.mark.parametrize("block_bytes,max_slice", [(1 << 16, None), (300, None), (4096, 500)])
def test_round_trip(tmp_path, block_bytes, max_slice):
    rng = np.random.RandomState(0)
    t = _tensors(rng)
    path = str(tmp_path / "model.ckpt")
    C.write_tf_v1_checkpoint(path, t, block_bytes=block_bytes, max_slice_elems=max_slice)
    got = C.read_tf_v1_checkpoint(path, verify_checksums=True)
    assert set(got) == set(t)
    for k in t:
        assert got[k].dtype == t[k].dtype and got[k].shape == t[k].shape
        np.testing.assert_array_equal(got[k], t[k])
    only = C.read_tf_v1_checkpoint(path, names=lambda n: n.startswith("InceptionV1/") and "Logits" not in n)
    assert sorted(only) == sorted(k for k in t if k.startswith("InceptionV1/") and "Logits" not in k)


def test_rejects_foreign_files_and_detects_corruption(tmp_path):
    bad = tmp_path / "x.ckpt"
    bad.write_bytes(b"\x00" * 100)
    with pytest.raises(ValueError, match="magic"):
        C.read_tf_v1_checkpoint(str(bad))
    path = str(tmp_path / "m.ckpt")
    C.write_tf_v1_checkpoint(path, {"a": np.arange(10, dtype=np.float32)})
    raw = bytearray(open(path, "rb").read())
    raw[10] ^= 0xFF
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        C.read_tf_v1_checkpoint(path, verify_checksums=True)


def test_prefix_compressed_block_by_hand():
    """A block written out byte by byte: shared-prefix key deltas, two restart points."""
    def entry(shared, delta, value):
        return bytes([shared, len(delta), len(value)]) + delta + value
    body = entry(0, b"apple", b"1") + entry(3, b"ly", b"22") + entry(0, b"banana", b"") + entry(6, b"s", b"4444")
    second_restart = len(entry(0, b"apple", b"1") + entry(3, b"ly", b"22"))
    block = body + struct.pack("<III", 0, second_restart, 2)
    assert list(C._block_entries(block)) == [(b"apple", b"1"), (b"apply", b"22"), (b"banana", b""), (b"bananas", b"4444")]


def test_snappy_blocks(tmp_path):
    # literal "abcd", copy(offset 4, length 8) with overlap semantics, long literal with a 1-byte length
    lit = bytes(range(70))
    stream = bytes([12 + len(lit)]) + bytes([(4 - 1) << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4]) + bytes([60 << 2, len(lit) - 1]) + lit
    assert C.snappy_uncompress(stream) == b"abcd" + b"abcdabcd" + lit
    # 2-byte-offset copy
    s2 = bytes([10]) + bytes([(5 - 1) << 2]) + b"hello" + bytes([((5 - 1) << 2) | 2, 5, 0])
    assert C.snappy_uncompress(s2) == b"hellohello"
    # a table whose only data block is snappy-compressed (all literals) is read like a raw one
    path = str(tmp_path / "m.ckpt")
    C.write_tf_v1_checkpoint(path, {"w": np.arange(6, dtype=np.float32).reshape(2, 3)})
    data = open(path, "rb").read()
    footer = data[-48:]
    (_, _), pos = C._handle(footer, 0)
    index, _ = C._handle(footer, pos)
    idx_entries = list(C._block_entries(C._read_block(data, index, True)))
    (off, size), _ = C._handle(idx_entries[0][1], 0)
    block = data[off:off + size]
    assert size - 1 < 60 * 256
    comp = C._varint(size) + bytes([61 << 2]) + struct.pack("<H", size - 1) + block      # one literal, 2-byte length
    new = bytearray(comp + b"\x01" + struct.pack("<I", masked_crc(comp + b"\x01")))
    meta_block = C._block([])
    def emit(buf, b):
        o = len(buf); buf += b + b"\x00" + struct.pack("<I", masked_crc(b + b"\x00")); return C._varint(o) + C._varint(len(b))
    mh = emit(new, meta_block)
    ih = emit(new, C._block([(idx_entries[0][0], C._varint(0) + C._varint(len(comp)))], restart_interval=1))
    new += mh + ih + b"\x00" * (40 - len(mh + ih)) + struct.pack("<Q", C.MAGIC)
    open(path, "wb").write(bytes(new))
    got = C.read_tf_v1_checkpoint(path, verify_checksums=True)
    np.testing.assert_array_equal(got["w"], np.arange(6, dtype=np.float32).reshape(2, 3))


def test_get_init_fn_prefers_the_tf_checkpoint(tmp_path):
    """im_model.get_init_fn finds inception_v1.ckpt, skips the Logits scope (im_model.py:118-137)."""
    from tumblr_emotions_amd.image_model.im_model import get_init_fn
    rng = np.random.RandomState(1)
    t = _tensors(rng)
    C.write_tf_v1_checkpoint(str(tmp_path / "inception_v1.ckpt"), t)
    fn = get_init_fn(str(tmp_path))
    assert fn is not None and get_init_fn(str(tmp_path / "missing")) is None

    wanted = [k for k in t if k.startswith("InceptionV1/") and "Logits" not in k]

    class FakeStore:
        def tf_names(self):
            return wanted + ["InceptionV1/Logits/Conv2d_0c_1x1/weights", "W_fc"]

    class FakeNet:
        store = FakeStore()

        def load_state_dict(self, sd, strict=True):
            self.sd, self.strict = sd, strict
            return set(sd)
    net = FakeNet()
    fn(net)
    assert net.strict is False
    # a checkpoint that lacks one of the model's InceptionV1 variables must not warm-start silently
    FakeStore.tf_names = lambda self: wanted + ["InceptionV1/Mixed_9z/weights"]
    with pytest.raises(KeyError):
        fn(FakeNet())
    assert sorted(net.sd) == sorted(k for k in t if k.startswith("InceptionV1/") and "Logits" not in k)
    np.testing.assert_array_equal(net.sd["InceptionV1/Conv2d_1a_7x7/weights"], t["InceptionV1/Conv2d_1a_7x7/weights"])


def test_handmade_v1_checkpoint_fixture():
    """tests/golden/handmade_v1.ckpt: a complete table assembled byte by byte by tests/golden/make_handmade_fixtures.py
    (independent of checkpoint_tf.py): a snappy data block with real back-reference copies, restart interval 2 with
    shared key prefixes, separator index keys, SavedSlice fields in reverse order, unpacked float_val / int64_val,
    tensor_content, packed int_val with a negative, a full extent written as an empty message, the VersionDef and
    per-tensor slice lists real TF files carry, a rank-0 tensor."""
    path = os.path.join(os.path.dirname(__file__), "golden", "handmade_v1.ckpt")
    got = C.read_tf_v1_checkpoint(path, verify_checksums=True)
    assert sorted(got) == ["a/weights", "b/step", "c/idx"]
    np.testing.assert_array_equal(got["a/weights"], np.array([[1.5, -2.0, 3.25], [4.0, 5.0, -6.5]], np.float32))
    assert got["a/weights"].dtype == np.float32
    assert got["b/step"].shape == () and got["b/step"].dtype == np.int64 and int(got["b/step"]) == 123456789012
    np.testing.assert_array_equal(got["c/idx"], np.array([7, -3, 0, 2 ** 31 - 1], np.int32))
    data = open(path, "rb").read()
    footer = data[-48:]
    _, pos = C._handle(footer, 0)
    index, _ = C._handle(footer, pos)
    handles = [C._handle(v, 0)[0] for _, v in C._block_entries(C._read_block(data, index, True))]
    assert data[handles[0][0] + handles[0][1]] == 1 and data[handles[1][0] + handles[1][1]] == 0     # snappy, raw
    raw0 = C._read_block(data, handles[0], True)
    assert len(raw0) > handles[0][1]                                # the compressed block really is smaller
    only = C.read_tf_v1_checkpoint(path, names=lambda n: n.startswith("c/"))
    assert list(only) == ["c/idx"]


def test_protobuf_runtime_serialised_checkpoint_messages():
    """tests/golden/protobuf_v1.ckpt: a V1 checkpoint table whose VALUES are SavedTensorSlices messages serialised by the
    official protobuf runtime (descriptors declared from TensorFlow's published .proto files in
    tests/golden/make_protobuf_fixtures.py): meta with VersionDef and per-tensor slice lists, tensor_content, packed
    float_val / double_val / int_val (negatives) / int64_val, a tensor saved as two row slices, a rank-0 tensor, extents
    with and without a length.  read_tf_v1_checkpoint must return the arrays that went into the runtime, bit for bit."""
    import json
    here = os.path.join(os.path.dirname(__file__), "golden")
    fx = json.load(open(os.path.join(here, "protobuf_fixtures.json")))["checkpoint"]
    got = C.read_tf_v1_checkpoint(os.path.join(here, "protobuf_v1.ckpt"), verify_checksums=True)
    assert sorted(got) == sorted(fx) and len(fx) == 6
    for name, want in fx.items():
        arr = np.frombuffer(bytes.fromhex(want["hex"]), dtype=want["dtype"]).reshape(want["shape"])
        assert got[name].dtype == arr.dtype and got[name].shape == arr.shape, name
        assert got[name].tobytes() == arr.tobytes(), name
    assert got["Text/idx"].tolist() == [7, -3, 0, 2 ** 31 - 1, -2 ** 31]
    assert int(got["global_step"]) == 1234567890123
    # the model-variable filter of get_init_fn (image_model/im_model.py:118-137) applied to runtime-serialised names
    only = C.read_tf_v1_checkpoint(os.path.join(here, "protobuf_v1.ckpt"), names=lambda n: n.startswith("InceptionV1/"))
    assert sorted(only) == sorted(n for n in fx if n.startswith("InceptionV1/"))
