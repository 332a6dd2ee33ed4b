#!/usr/bin/env python
"""Hand-assembled format fixtures for SURVEY rows 8f-2 (TFRecord / tf.train.Example) and 8f-4 (TensorFlow V1
"tensor slice" checkpoint), written byte by byte from the published format descriptions -- protobuf wire format,
tensorflow/core/lib/io/record_writer.h (record framing, masked CRC-32C), tensorflow/core/lib/io/format.h +
block_builder.cc (table blocks), saved_tensor_slice.proto / tensor.proto / tensor_slice.proto, and the snappy
format description -- WITHOUT importing anything from tumblr_emotions_amd, and deliberately in encodings this
build's own writers never produce (see the inline comments).  They are not TensorFlow-written files (TensorFlow is
not installable here), so they pin the readers against the published specs, not against TensorFlow itself:
"parity unpinned" stays in the docs until a real inception_v1.ckpt / TF-written TFRecord can be read.

    python tests/golden/make_handmade_fixtures.py      # rewrites handmade_examples.tfrecord, handmade_v1.ckpt
"""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))


# ---- primitives, restated independently of the package ------------------------------------------------------
def crc32c_bitwise(data):
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), one bit at a time."""
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def masked(data):
    """record_writer.h / crc32c.h Mask(): rotate right by 15 bits, add 0xa282ead8."""
    c = crc32c_bitwise(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def key(field, wire_type):
    return varint((field << 3) | wire_type)


def ld(field, payload):            # length-delimited field
    return key(field, 2) + varint(len(payload)) + payload


# ---- 8f-2: two framed tf.train.Example records ------------------------------------------------------------------
def example_a():
    """Feature map in NON-sorted order; Int64List and FloatList UNPACKED (one key per element, legal proto2/3
    wire data every parser must accept); a map entry with the value BEFORE the key; a 10-byte negative varint;
    a two-byte length prefix; an unknown field inside a Feature and one inside the Example (ignored by parsers)."""
    text = b"".join(key(1, 0) + varint(v) for v in (5, 300, 70000, -1))           # Int64List.value, unpacked
    f_text = ld(3, text)
    f_image = ld(1, ld(1, b"\x89PNG-not-really") + ld(1, b"second"))              # BytesList with two values
    f_seq = ld(3, ld(1, varint(300) + varint(2)))                                 # packed, multi-byte varint
    f_w = ld(2, key(1, 5) + struct.pack("<f", 0.5) + key(1, 5) + struct.pack("<f", -2.25))   # FloatList unpacked
    f_fmt = key(15, 0) + varint(7) + ld(1, ld(1, b"png"))                         # unknown field 15 first
    long_name = b"k" * 200                                                        # entry length needs 2 varint bytes
    f_long = ld(3, ld(1, varint(1)))
    entries = (ld(1, ld(1, b"text") + ld(2, f_text)) +
               ld(1, ld(1, b"image/encoded") + ld(2, f_image)) +
               ld(1, ld(2, f_fmt) + ld(1, b"image/format")) +                     # value before key
               ld(1, ld(1, b"seq_len") + ld(2, f_seq)) +
               ld(1, ld(1, b"weights") + ld(2, f_w)) +
               ld(1, ld(1, long_name) + ld(2, f_long)))
    return key(9, 0) + varint(1) + ld(1, entries)                                 # unknown Example field 9 first


def frame(data):
    head = struct.pack("<Q", len(data))
    return head + struct.pack("<I", masked(head)) + data + struct.pack("<I", masked(data))


def write_examples(path):
    with open(path, "wb") as f:
        f.write(frame(example_a()))
        f.write(frame(b""))                     # an Example with no features at all: zero-length record


# ---- snappy compressor (greedy, 4-byte hash matches, 2-byte-offset copies) -------------------------------------
def snappy_compress(data):
    out = bytearray(varint(len(data)))
    table, i, lit_start = {}, 0, 0

    def flush_literal(end):
        n = end - lit_start
        if n <= 0:
            return
        if n <= 60:
            out.append((n - 1) << 2)
        elif n <= 256:
            out.extend(bytes([60 << 2, n - 1]))
        else:
            out.extend(bytes([61 << 2]) + struct.pack("<H", n - 1))
        out.extend(data[lit_start:end])

    while i + 4 <= len(data):
        h = data[i:i + 4]
        j = table.get(h)
        table[h] = i
        if j is not None and i - j < 65536:
            ln = 4
            while i + ln < len(data) and ln < 64 and data[j + ln] == data[i + ln]:
                ln += 1
            flush_literal(i)
            out.extend(bytes([((ln - 1) << 2) | 2]) + struct.pack("<H", i - j))   # copy with 2-byte offset
            i += ln
            lit_start = i
        else:
            i += 1
    flush_literal(len(data))
    return bytes(out)


# ---- 8f-4: a complete V1 checkpoint table ------------------------------------------------------------------------
def shape_proto(dims):
    return b"".join(ld(2, key(1, 0) + varint(d)) for d in dims)


def block(entries, restart_interval):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        prev = k
    for r in restarts or [0]:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts or [0]))
    return bytes(out)


def write_checkpoint(path):
    DT_FLOAT, DT_INT32, DT_INT64 = 1, 3, 9
    # SavedTensorSliceMeta: field order type(3) before shape(2); the per-tensor `slice` list (4) present;
    # a VersionDef (field 2 of the meta message) as real TF files carry
    def meta(name, dims, dtype, nslices):
        slices = b"".join(ld(4, ld(1, b"")) for _ in range(nslices))
        return ld(1, ld(1, name) + key(3, 0) + varint(dtype) + ld(2, shape_proto(dims)) + slices)
    metas = (meta(b"a/weights", (2, 3), DT_FLOAT, 2) + meta(b"b/step", (), DT_INT64, 1) + meta(b"c/idx", (4,), DT_INT32, 1))
    meta_value = ld(1, metas + ld(2, key(1, 0) + varint(21)))
    # a/weights rows [0:1]: extent {start 0, length 1} + a FULL extent (empty message = whole dimension);
    # payload as UNPACKED float_val (one fixed32 per element); fields of SavedSlice in reverse order (3, 2, 1)
    row0 = b"".join(key(5, 5) + struct.pack("<f", v) for v in (1.5, -2.0, 3.25))
    s0 = ld(2, ld(3, key(1, 0) + varint(DT_FLOAT) + ld(2, shape_proto((1, 3))) + row0) +
            ld(2, ld(1, key(1, 0) + varint(0) + key(2, 0) + varint(1)) + ld(1, b"")) +
            ld(1, b"a/weights"))
    # a/weights rows [1:2]: tensor_content (field 4), the encoding TF uses for large tensors
    s1 = ld(2, ld(1, b"a/weights") +
            ld(2, ld(1, key(1, 0) + varint(1) + key(2, 0) + varint(1)) + ld(1, b"")) +
            ld(3, key(1, 0) + varint(DT_FLOAT) + ld(4, struct.pack("<3f", 4.0, 5.0, -6.5))))
    # b/step: rank-0 int64, unpacked int64_val, no extents at all
    s2 = ld(2, ld(1, b"b/step") + ld(2, b"") + ld(3, key(1, 0) + varint(DT_INT64) + key(10, 0) + varint(123456789012)))
    # c/idx: packed int_val with a negative element (10-byte varint)
    s3 = ld(2, ld(1, b"c/idx") + ld(2, ld(1, b"")) +
            ld(3, key(1, 0) + varint(DT_INT32) + ld(7, varint(7) + varint(-3) + varint(0) + varint(2 ** 31 - 1))))
    # data keys only need to sort: arbitrary OrderedCode-like bytes sharing long prefixes
    block0 = block([(b"", meta_value), (b"\x00a/weights\x00\x01\x00", s0), (b"\x00a/weights\x00\x01\x01", s1)], restart_interval=2)
    block1 = block([(b"\x00b/step\x00\x01", s2), (b"\x00c/idx\x00\x01", s3)], restart_interval=16)
    out = bytearray()

    def emit(raw, compress):
        off = len(out)
        body = snappy_compress(raw) if compress else raw
        typ = b"\x01" if compress else b"\x00"
        out.extend(body + typ + struct.pack("<I", masked(body + typ)))
        return varint(off) + varint(len(body))

    h0 = emit(block0, True)                       # snappy block with real back-reference copies
    h1 = emit(block1, False)
    hm = emit(block([], 16), False)
    # index keys are separators >= last key of the block, not the last key itself
    hi = emit(block([(b"\x00a/weights\x00\x02", h0), (b"\x00d", h1)], restart_interval=1), False)
    footer = hm + hi
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57))
    with open(path, "wb") as f:
        f.write(bytes(out))


if __name__ == "__main__":
    write_examples(os.path.join(HERE, "handmade_examples.tfrecord"))
    write_checkpoint(os.path.join(HERE, "handmade_v1.ckpt"))
    for n in ("handmade_examples.tfrecord", "handmade_v1.ckpt"):
        print(n, os.path.getsize(os.path.join(HERE, n)), "bytes")
