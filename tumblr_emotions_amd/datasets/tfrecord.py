"""TFRecord framing and tf.train.Example wire format without TensorFlow.

The reference stores its dataset as sharded TFRecord files of tf.train.Example protos
(datasets/convert_images_tfrecords.py:110-113,180-230; schema datasets/dataset_utils.py:65-76) and
reads them back through slim's TFExampleDecoder (datasets/convert_to_dataset.py:148-170).  TensorFlow
is not available here, so both directions are restated from the published formats [TF-sem]:

  record  = uint64 length | uint32 masked_crc32c(length) | bytes data | uint32 masked_crc32c(data)
  masked  = ((crc >> 15) | (crc << 17)) + 0xa282ead8   (mod 2^32), crc = CRC-32C (Castagnoli)
  Example = { 1: Features { 1: repeated MapEntry { 1: key string, 2: Feature } } }
  Feature = oneof { 1: BytesList{1: repeated bytes}, 2: FloatList{1: packed float}, 3: Int64List{1: packed varint} }

Pinned by the CRC-32C check values (0xE3069283 for b"123456789", the RFC 3720 B.4 vectors), by write/read round
trips, and by tests/golden/handmade_examples.tfrecord -- records assembled byte by byte from the published
specs by an independent script, in encodings this module's writer never emits (unsorted map, unpacked lists,
value-before-key entries, unknown fields, zero-length record).  There is no TensorFlow-WRITTEN file in the
reference tree to pin the framing against: parity unpinned.
"""
import struct

import numpy as np

_CRC_TABLE = None


def _table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        poly = 0x82F63B78
        t = np.zeros(256, dtype=np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ poly if c & 1 else c >> 1
            t[i] = c
        _CRC_TABLE = [int(v) for v in t]
    return _CRC_TABLE


def crc32c(data):
    t = _table()
    c = 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def write_records(path, records):
    with open(path, "wb") as f:
        for r in records:
            head = struct.pack("<Q", len(r))
            f.write(head + struct.pack("<I", masked_crc(head)) + r + struct.pack("<I", masked_crc(r)))


def read_records(path, verify=False):
    with open(path, "rb") as f:
        while True:
            head = f.read(8)
            if not head:
                return
            if len(head) < 8:
                raise IOError("truncated TFRecord header in %s" % path)
            (n,) = struct.unpack("<Q", head)
            (hcrc,) = struct.unpack("<I", f.read(4))
            data = f.read(n)
            (dcrc,) = struct.unpack("<I", f.read(4))
            if len(data) < n:
                raise IOError("truncated TFRecord in %s" % path)
            if verify and (hcrc != masked_crc(head) or dcrc != masked_crc(data)):
                raise IOError("corrupt TFRecord (crc mismatch) in %s" % path)
            yield data


# ---- protobuf wire format (just what tf.train.Example needs) ----------------------------------------
def _varint(n):
    n &= (1 << 64) - 1            # int64 two's complement, as protobuf encodes negative varints
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _len_field(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _fields(buf):
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _read_varint(buf, pos)
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, val


def encode_example(features):
    """features: {name: bytes | str | int | float | list of one of those} -> serialized tf.train.Example."""
    entries = b""
    for name in sorted(features):
        v = features[name]
        vals = list(v) if isinstance(v, (list, tuple, np.ndarray)) else [v]
        if vals and isinstance(vals[0], (bytes, str)):
            body = b"".join(_len_field(1, x if isinstance(x, bytes) else x.encode()) for x in vals)
            feat = _len_field(1, body)
        elif vals and isinstance(vals[0], (float, np.floating)):
            feat = _len_field(2, _len_field(1, struct.pack("<%df" % len(vals), *vals)))
        else:
            feat = _len_field(3, _len_field(1, b"".join(_varint(int(x)) for x in vals)))
        entries += _len_field(1, _len_field(1, name.encode()) + _len_field(2, feat))
    return _len_field(1, entries)


def decode_example(buf):
    """serialized tf.train.Example -> {name: list of bytes | floats | ints}."""
    out = {}
    for f1, w1, features in _fields(buf):
        if f1 != 1 or w1 != 2:
            continue
        for f2, w2, entry in _fields(features):
            if f2 != 1 or w2 != 2:
                continue
            key, feat = None, b""
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    key = bytes(v).decode()
                elif f3 == 2:
                    feat = v
            vals = []
            for kind, kwt, lst in _fields(feat):
                if kind not in (1, 2, 3) or kwt != 2:        # unknown fields are skipped, as protobuf parsers do
                    continue
                for f5, wt, v in _fields(lst):
                    if f5 != 1:
                        continue
                    if kind == 1:
                        vals.append(bytes(v))
                    elif kind == 2:
                        vals.extend(struct.unpack("<%df" % (len(v) // 4), v) if wt == 2 else struct.unpack("<f", v))
                    elif kind == 3:
                        if wt == 2:          # packed
                            p = 0
                            while p < len(v):
                                x, p = _read_varint(v, p)
                                vals.append(x - (1 << 64) if x >> 63 else x)
                        else:
                            vals.append(v - (1 << 64) if v >> 63 else v)
            out[key] = vals
    return out
