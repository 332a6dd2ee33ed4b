#!/usr/bin/env python
"""Format fixtures for SURVEY rows 8f-2 / 8f-4 serialised by the OFFICIAL protobuf runtime (google.protobuf, installed in
the build container) instead of by hand: an independent third-party ENCODER for the messages the readers decode.

The message types are declared here from TensorFlow's published .proto files (tensorflow/core/example/example.proto,
feature.proto; tensorflow/core/util/saved_tensor_slice.proto, tensorflow/core/framework/tensor.proto, tensor_shape.proto,
tensor_slice.proto, types.proto, versions.proto) through descriptor_pb2 -- field numbers, types, labels, packed options and
the map entry of `Features.feature` as published -- and the runtime does the encoding.  Nothing is imported from
tumblr_emotions_amd.  What is serialised:

  * protobuf_examples.tfrecord: tf.train.Example records with the dataset schema of
    /root/reference/datasets/convert_to_dataset.py:148-161 (image/encoded, image/format, image/class/label, text[50],
    seq_len, post_id, day) as datasets/dataset_utils.py:65-76 builds them, incl. a negative int64, an empty text, a
    multi-value bytes list, floats, and a record carrying UNKNOWN fields (a runtime-declared superset message);
    the record framing (length, masked CRC-32C) is make_handmade_fixtures.py's independent implementation.
  * protobuf_v1.ckpt: a V1 "tensor slice" checkpoint whose table VALUES are runtime-serialised SavedTensorSlices messages
    (meta with VersionDef and per-tensor slice lists; data slices with tensor_content, packed float_val / double_val /
    int_val / int64_val, a sliced tensor, a rank-0 tensor, extents with and without length); the table framing around them
    (prefix-compressed blocks, restart arrays, snappy, footer) is again make_handmade_fixtures.py's.
  * protobuf_fixtures.json: what the readers must return (written from the Python values that went INTO the runtime).

These are still not TensorFlow-written files (TensorFlow cannot be installed here): "parity unpinned" stays for the two
rows; but the codecs are no longer checked only against bytes assembled by the same author.

    python tests/golden/make_protobuf_fixtures.py      # rewrites the three files (deterministic serialisation)
"""
import json
import os
import struct
import sys

from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_handmade_fixtures as H      # framing only: frame(), block(), snappy_compress(), masked(), varint()

F = descriptor_pb2.FieldDescriptorProto
OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED


def _field(msg, name, number, ftype, label=OPT, type_name=None, packed=None, oneof=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name
    if packed is not None:
        f.options.packed = packed
    if oneof is not None:
        f.oneof_index = oneof
    return f


def build_pool():
    pool = descriptor_pool.DescriptorPool()
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "ds_tf_formats.proto", "tensorflow", "proto3"

    # ---- feature.proto / example.proto ---------------------------------------------------------------------------------
    m = fd.message_type.add(); m.name = "BytesList"
    _field(m, "value", 1, F.TYPE_BYTES, REP)
    m = fd.message_type.add(); m.name = "FloatList"
    _field(m, "value", 1, F.TYPE_FLOAT, REP, packed=True)
    m = fd.message_type.add(); m.name = "Int64List"
    _field(m, "value", 1, F.TYPE_INT64, REP, packed=True)
    m = fd.message_type.add(); m.name = "Feature"
    m.oneof_decl.add().name = "kind"
    _field(m, "bytes_list", 1, F.TYPE_MESSAGE, type_name=".tensorflow.BytesList", oneof=0)
    _field(m, "float_list", 2, F.TYPE_MESSAGE, type_name=".tensorflow.FloatList", oneof=0)
    _field(m, "int64_list", 3, F.TYPE_MESSAGE, type_name=".tensorflow.Int64List", oneof=0)
    m = fd.message_type.add(); m.name = "Features"
    e = m.nested_type.add(); e.name = "FeatureEntry"; e.options.map_entry = True
    _field(e, "key", 1, F.TYPE_STRING)
    _field(e, "value", 2, F.TYPE_MESSAGE, type_name=".tensorflow.Feature")
    _field(m, "feature", 1, F.TYPE_MESSAGE, REP, type_name=".tensorflow.Features.FeatureEntry")
    m = fd.message_type.add(); m.name = "Example"
    _field(m, "features", 1, F.TYPE_MESSAGE, type_name=".tensorflow.Features")
    # a SUPERSET of Example / Feature with extra fields: what a newer writer could emit; readers must skip them
    m = fd.message_type.add(); m.name = "FeaturePlus"
    _field(m, "bytes_list", 1, F.TYPE_MESSAGE, type_name=".tensorflow.BytesList")
    _field(m, "int64_list", 3, F.TYPE_MESSAGE, type_name=".tensorflow.Int64List")
    _field(m, "note", 12, F.TYPE_STRING)
    _field(m, "weight", 13, F.TYPE_DOUBLE)
    m = fd.message_type.add(); m.name = "FeaturesPlus"
    e = m.nested_type.add(); e.name = "FeatureEntry"; e.options.map_entry = True
    _field(e, "key", 1, F.TYPE_STRING)
    _field(e, "value", 2, F.TYPE_MESSAGE, type_name=".tensorflow.FeaturePlus")
    _field(m, "feature", 1, F.TYPE_MESSAGE, REP, type_name=".tensorflow.FeaturesPlus.FeatureEntry")
    m = fd.message_type.add(); m.name = "ExamplePlus"
    _field(m, "features", 1, F.TYPE_MESSAGE, type_name=".tensorflow.FeaturesPlus")
    _field(m, "source", 7, F.TYPE_STRING)
    _field(m, "serial", 9, F.TYPE_FIXED64)

    # ---- tensor_shape.proto / tensor_slice.proto / tensor.proto / versions.proto / saved_tensor_slice.proto -------------
    m = fd.message_type.add(); m.name = "TensorShapeProto"
    d = m.nested_type.add(); d.name = "Dim"
    _field(d, "size", 1, F.TYPE_INT64)
    _field(d, "name", 2, F.TYPE_STRING)
    _field(m, "dim", 2, F.TYPE_MESSAGE, REP, type_name=".tensorflow.TensorShapeProto.Dim")
    _field(m, "unknown_rank", 3, F.TYPE_BOOL)
    m = fd.message_type.add(); m.name = "TensorSliceProto"
    x = m.nested_type.add(); x.name = "Extent"
    x.oneof_decl.add().name = "has_length"
    _field(x, "start", 1, F.TYPE_INT64)
    _field(x, "length", 2, F.TYPE_INT64, oneof=0)
    _field(m, "extent", 1, F.TYPE_MESSAGE, REP, type_name=".tensorflow.TensorSliceProto.Extent")
    m = fd.message_type.add(); m.name = "TensorProto"
    _field(m, "dtype", 1, F.TYPE_INT32)                      # enum DataType on the wire = varint
    _field(m, "tensor_shape", 2, F.TYPE_MESSAGE, type_name=".tensorflow.TensorShapeProto")
    _field(m, "version_number", 3, F.TYPE_INT32)
    _field(m, "tensor_content", 4, F.TYPE_BYTES)
    _field(m, "float_val", 5, F.TYPE_FLOAT, REP, packed=True)
    _field(m, "double_val", 6, F.TYPE_DOUBLE, REP, packed=True)
    _field(m, "int_val", 7, F.TYPE_INT32, REP, packed=True)
    _field(m, "string_val", 8, F.TYPE_BYTES, REP)
    _field(m, "int64_val", 10, F.TYPE_INT64, REP, packed=True)
    m = fd.message_type.add(); m.name = "VersionDef"
    _field(m, "producer", 1, F.TYPE_INT32)
    _field(m, "min_consumer", 2, F.TYPE_INT32)
    _field(m, "bad_consumers", 3, F.TYPE_INT32, REP, packed=True)
    m = fd.message_type.add(); m.name = "SavedSliceMeta"
    _field(m, "name", 1, F.TYPE_STRING)
    _field(m, "shape", 2, F.TYPE_MESSAGE, type_name=".tensorflow.TensorShapeProto")
    _field(m, "type", 3, F.TYPE_INT32)
    _field(m, "slice", 4, F.TYPE_MESSAGE, REP, type_name=".tensorflow.TensorSliceProto")
    m = fd.message_type.add(); m.name = "SavedTensorSliceMeta"
    _field(m, "tensor", 1, F.TYPE_MESSAGE, REP, type_name=".tensorflow.SavedSliceMeta")
    _field(m, "versions", 2, F.TYPE_MESSAGE, type_name=".tensorflow.VersionDef")
    m = fd.message_type.add(); m.name = "SavedSlice"
    _field(m, "name", 1, F.TYPE_STRING)
    _field(m, "slice", 2, F.TYPE_MESSAGE, type_name=".tensorflow.TensorSliceProto")
    _field(m, "data", 3, F.TYPE_MESSAGE, type_name=".tensorflow.TensorProto")
    m = fd.message_type.add(); m.name = "SavedTensorSlices"
    _field(m, "meta", 1, F.TYPE_MESSAGE, type_name=".tensorflow.SavedTensorSliceMeta")
    _field(m, "data", 2, F.TYPE_MESSAGE, type_name=".tensorflow.SavedSlice")
    pool.Add(fd)
    return pool


POOL = build_pool()


def cls(name):
    return message_factory.GetMessageClass(POOL.FindMessageTypeByName("tensorflow." + name))


# ---- 8f-2 -----------------------------------------------------------------------------------------------------------------
POST_SIZE = 50      # datasets/convert_to_dataset.py:16 (_POST_SIZE)


def dataset_example(jpeg, label, text, seq_len, post_id, day):
    """datasets/dataset_utils.py:65-76 image_to_tfexample(): the seven features of the converted dataset."""
    ex = cls("Example")()
    f = ex.features.feature
    f["image/encoded"].bytes_list.value.append(jpeg)
    f["image/format"].bytes_list.value.append(b"jpg")
    f["image/class/label"].int64_list.value.append(label)
    f["text"].int64_list.value.extend(text)
    f["seq_len"].int64_list.value.append(seq_len)
    f["post_id"].int64_list.value.append(post_id)
    f["day"].int64_list.value.append(day)
    return ex


def examples():
    recs, expect = [], []
    jpeg0 = bytes(range(256)) * 3 + b"\xff\xd9"
    text0 = [(7 * i) % 10001 for i in range(POST_SIZE)]
    ex = dataset_example(jpeg0, 11, text0, 37, 163542871234, 4)
    recs.append(ex.SerializeToString(deterministic=True))
    expect.append({"image/encoded": [jpeg0.hex()], "image/format": [b"jpg".hex()], "image/class/label": [11], "text": text0,
                   "seq_len": [37], "post_id": [163542871234], "day": [4]})
    # negative int64 (10-byte varint inside a packed run), the unknown-word id 400000 (3-byte varints), an empty jpeg
    text1 = [400000] * 6 + [-1] + [0] * (POST_SIZE - 7)
    ex = dataset_example(b"", 0, text1, 6, -5, 0)
    recs.append(ex.SerializeToString(deterministic=True))
    expect.append({"image/encoded": [b"".hex()], "image/format": [b"jpg".hex()], "image/class/label": [0], "text": text1, "seq_len": [6],
                   "post_id": [-5], "day": [0]})
    # floats, a two-value bytes list, an empty int64 list, a feature with no kind set at all
    ex = cls("Example")()
    f = ex.features.feature
    f["weights"].float_list.value.extend([0.5, -2.25, 3.0e38, 1.0e-42])
    f["names"].bytes_list.value.extend([b"first", b"", b"third \x00 value"])
    f["empty"].int64_list.SetInParent()
    f["unset"].SetInParent()
    recs.append(ex.SerializeToString(deterministic=True))
    expect.append({"weights": [struct.unpack("<f", struct.pack("<f", v))[0] for v in (0.5, -2.25, 3.0e38, 1.0e-42)],
                   "names": [b"first".hex(), b"".hex(), b"third \x00 value".hex()], "empty": [], "unset": []})
    # a record from a "newer writer": extra fields in Feature and in Example that a reader must skip
    ex = cls("ExamplePlus")()
    ex.source, ex.serial = "scraper-v2", 0xDEADBEEFCAFE
    f = ex.features.feature
    f["image/format"].bytes_list.value.append(b"png")
    f["image/format"].note = "re-encoded"
    f["seq_len"].int64_list.value.append(19)
    f["seq_len"].weight = 0.125
    recs.append(ex.SerializeToString(deterministic=True))
    expect.append({"image/format": [b"png".hex()], "seq_len": [19]})
    return recs, expect


# ---- 8f-4 -----------------------------------------------------------------------------------------------------------------
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9


def shape_msg(msg, dims):
    for d in dims:
        msg.dim.add().size = d


def checkpoint_messages():
    """[(table key, serialised SavedTensorSlices)] + the expected tensors."""
    import numpy as np
    rng = np.random.RandomState(5)
    conv = rng.normal(size=(3, 3, 4, 6)).astype("<f4")                       # InceptionV1-style conv weights: tensor_content
    beta = rng.normal(size=(6,)).astype("<f4")                               # packed float_val
    mvar = np.abs(rng.normal(size=(6,))).astype("<f8")                       # packed double_val
    step = np.array(1234567890123, dtype="<i8")                              # rank-0 int64
    idx = np.array([7, -3, 0, 2 ** 31 - 1, -2 ** 31], dtype="<i4")           # packed int_val, negatives
    emb = rng.normal(size=(5, 3)).astype("<f4")                              # saved as TWO row slices
    tensors = [("InceptionV1/Conv2d_2c_3x3/weights", conv, DT_FLOAT), ("InceptionV1/Conv2d_2c_3x3/BatchNorm/beta", beta, DT_FLOAT),
               ("InceptionV1/Conv2d_2c_3x3/BatchNorm/moving_variance", mvar, DT_DOUBLE), ("global_step", step, DT_INT64),
               ("Text/idx", idx, DT_INT32), ("Text/W_embedding", emb, DT_FLOAT)]
    S = cls("SavedTensorSlices")
    meta = S()
    meta.meta.versions.producer, meta.meta.versions.min_consumer = 21, 0
    for name, arr, dt in tensors:
        t = meta.meta.tensor.add()
        t.name, t.type = name, dt
        shape_msg(t.shape, arr.shape)
        nsl = 2 if name == "Text/W_embedding" else 1
        for s in range(nsl):
            sl = t.slice.add()
            for d in range(arr.ndim):
                e = sl.extent.add()
                if nsl == 2 and d == 0:
                    e.start, e.length = (0, 2) if s == 0 else (2, 3)
    entries = [(b"", meta.SerializeToString(deterministic=True))]

    def data(name, arr, dt, extents, payload):
        m = S()
        m.data.name = name
        for (start, length) in extents:
            e = m.data.slice.extent.add()
            if start is not None:
                e.start = start
            if length is not None:
                e.length = length
        m.data.data.dtype = dt
        shape_msg(m.data.data.tensor_shape, arr.shape)
        payload(m.data.data, arr)
        return m.SerializeToString(deterministic=True)

    full = lambda a: [(None, None)] * a.ndim
    entries.append((b"\x00InceptionV1/Conv2d_2c_3x3/weights\x00\x01", data(tensors[0][0], conv, DT_FLOAT, full(conv),
                   lambda t, a: setattr(t, "tensor_content", a.tobytes()))))
    entries.append((b"\x00InceptionV1/Conv2d_2c_3x3/BatchNorm/beta\x00\x01", data(tensors[1][0], beta, DT_FLOAT, full(beta),
                   lambda t, a: t.float_val.extend(float(v) for v in a))))
    entries.append((b"\x00InceptionV1/Conv2d_2c_3x3/BatchNorm/moving_variance\x00\x01", data(tensors[2][0], mvar, DT_DOUBLE, full(mvar),
                   lambda t, a: t.double_val.extend(float(v) for v in a))))
    entries.append((b"\x00global_step\x00\x01", data(tensors[3][0], step, DT_INT64, [],
                   lambda t, a: t.int64_val.append(int(a)))))
    entries.append((b"\x00Text/idx\x00\x01", data(tensors[4][0], idx, DT_INT32, full(idx),
                   lambda t, a: t.int_val.extend(int(v) for v in a))))
    entries.append((b"\x00Text/W_embedding\x00\x01\x00", data(tensors[5][0], emb[0:2], DT_FLOAT, [(0, 2), (None, None)],
                   lambda t, a: t.float_val.extend(float(v) for v in a.ravel()))))
    entries.append((b"\x00Text/W_embedding\x00\x01\x02", data(tensors[5][0], emb[2:5], DT_FLOAT, [(2, 3), (None, None)],
                   lambda t, a: setattr(t, "tensor_content", a.tobytes()))))
    expect = {name: {"shape": list(arr.shape), "dtype": str(arr.dtype), "hex": arr.tobytes().hex()} for name, arr, _ in tensors}
    return entries, expect


def write_table(path, entries):
    """The table around the runtime-serialised values: make_handmade_fixtures.py's independent block / footer writer."""
    entries = [entries[0]] + sorted(entries[1:])
    blocks = [entries[:3], entries[3:6], entries[6:]]
    out = bytearray()

    def emit(raw, compress):
        off = len(out)
        body = H.snappy_compress(raw) if compress else raw
        typ = b"\x01" if compress else b"\x00"
        out.extend(body + typ + struct.pack("<I", H.masked(body + typ)))
        return H.varint(off) + H.varint(len(body))

    handles = []
    for i, b in enumerate(blocks):
        handles.append((b[-1][0] + b"\xff", emit(H.block(b, restart_interval=(1, 2, 16)[i]), compress=(i != 1))))
    hm = emit(H.block([], 16), False)
    hi = emit(H.block(handles, restart_interval=1), False)
    footer = hm + hi
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57))
    with open(path, "wb") as f:
        f.write(bytes(out))


if __name__ == "__main__":
    recs, expect_ex = examples()
    with open(os.path.join(HERE, "protobuf_examples.tfrecord"), "wb") as f:
        for r in recs:
            f.write(H.frame(r))
    entries, expect_ck = checkpoint_messages()
    write_table(os.path.join(HERE, "protobuf_v1.ckpt"), entries)
    import google.protobuf
    with open(os.path.join(HERE, "protobuf_fixtures.json"), "w") as f:
        json.dump({"protobuf_runtime": google.protobuf.__version__, "examples": expect_ex, "checkpoint": expect_ck,
                   "messages_hex": {k.hex(): v.hex() for k, v in entries}}, f, indent=1, sort_keys=True)
    for n in ("protobuf_examples.tfrecord", "protobuf_v1.ckpt", "protobuf_fixtures.json"):
        print(n, os.path.getsize(os.path.join(HERE, n)), "bytes")
