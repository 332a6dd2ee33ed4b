"""TensorFlow V1 ("tensor slice") checkpoint reader -- the format of slim's `inception_v1.ckpt`, which the
reference warm-starts from (image_model/im_model.py:118-137; download URL in parallel_computing/job_train.py:17)
-- without TensorFlow.

Restated from the published formats [TF-sem]; NOT validated against a TensorFlow-written file (none exists in
the reference tree, TensorFlow is not installable here, there is no network): the tests pin it only against
this module's own writer and against hand-assembled blocks.  Parity unpinned.

  file     = LevelDB-style sorted table (tensorflow/core/lib/io/table*, format.h):
             data blocks | metaindex block | index block | 48-byte footer
  footer   = BlockHandle(metaindex) BlockHandle(index) zero padding to 40 bytes | magic 0xdb4775248b80fb57 (LE)
  handle   = varint64 offset, varint64 size (size excludes the 5-byte block trailer)
  block    = entries | uint32 restart offsets[] | uint32 num_restarts ; trailer = type byte (0 raw, 1 snappy)
             + masked CRC-32C of (block, type)
  entry    = varint32 shared_key_bytes, varint32 unshared_key_bytes, varint32 value_bytes, key delta, value
  values   = SavedTensorSlices protos (tensorflow/core/util/saved_tensor_slice.proto):
             key ""  -> { 1: SavedTensorSliceMeta { 1: repeated SavedSliceMeta {1 name, 2 shape, 3 dtype, 4 slices} } }
             others  -> { 2: SavedSlice { 1 name, 2 TensorSliceProto, 3 TensorProto } }
  TensorProto      = { 1 dtype, 2 shape, 4 tensor_content, 5 float_val, 6 double_val, 7 int_val, 10 int64_val }
  TensorShapeProto = { 2: repeated Dim { 1 size } } ; TensorSliceProto = { 1: repeated Extent { 1 start, 2 length } }

Keys of the data entries (an OrderedCode of name and slice) are never decoded: every value carries its
tensor's name.
"""
import struct

import numpy as np

from .datasets.tfrecord import _fields, _len_field, _read_varint, _varint, masked_crc

MAGIC = 0xdb4775248b80fb57
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9
_NP = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_INT64: np.int64}


# ---- snappy (raw block format), in case a writer enabled block compression ---------------------------------------
def snappy_uncompress(buf):
    n, pos = _read_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], "little")
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy block")
        for _ in range(ln):                             # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


# ---- table ---------------------------------------------------------------------------------------------------
def _handle(buf, pos):
    off, pos = _read_varint(buf, pos)
    size, pos = _read_varint(buf, pos)
    return (off, size), pos


def _read_block(data, handle, verify):
    off, size = handle
    raw, typ = data[off:off + size], data[off + size]
    if verify:
        want = struct.unpack("<I", data[off + size + 1:off + size + 5])[0]
        if masked_crc(data[off:off + size + 1]) != want:
            raise ValueError("block checksum mismatch at offset %d" % off)
    if typ == 1:
        raw = snappy_uncompress(raw)
    elif typ != 0:
        raise ValueError("unknown block compression type %d" % typ)
    return raw


def _block_entries(block):
    n_restarts = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _read_varint(block, pos)
        unshared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + block[pos:pos + unshared]
        pos += unshared
        yield key, block[pos:pos + vlen]
        pos += vlen


def table_entries(data, verify=False):
    """(key, value) pairs of a table file, in key order."""
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != MAGIC:
        raise ValueError("not a TensorFlow V1 checkpoint / LevelDB table (bad magic number)")
    footer = data[-48:]
    _, pos = _handle(footer, 0)                          # metaindex: unused
    index, _ = _handle(footer, pos)
    for _, hv in _block_entries(_read_block(data, index, verify)):
        h, _ = _handle(hv, 0)
        for kv in _block_entries(_read_block(data, h, verify)):
            yield kv


# ---- protos ---------------------------------------------------------------------------------------------------
def _shape(buf):
    dims = []
    for f, w, v in _fields(buf):
        if f == 2:
            size = 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    size = v2
            dims.append(size if size < (1 << 63) else size - (1 << 64))
    return tuple(dims)


def _extents(buf, shape):
    """TensorSliceProto -> list of python slices (an extent without a length spans the whole dimension)."""
    out = []
    for f, w, v in _fields(buf):
        if f == 1:
            start, length = 0, None
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    start = v2
                elif f2 == 2:
                    length = v2
            d = len(out)
            out.append(slice(start, shape[d] if length is None else start + length))
    while len(out) < len(shape):
        out.append(slice(0, shape[len(out)]))
    return tuple(out)


def _packed_varints(v, signed_bits):
    vals, pos = [], 0
    while pos < len(v):
        x, pos = _read_varint(v, pos)
        if x >= 1 << (signed_bits - 1) and signed_bits == 64:
            x -= 1 << 64
        vals.append(x)
    return vals


def _tensor_values(buf, dtype):
    """Flat numpy array of a TensorProto's payload (typed repeated field or tensor_content)."""
    np_t = _NP[dtype]
    chunks, content = [], None
    for f, w, v in _fields(buf):
        if f == 4:
            content = np.frombuffer(v, dtype=np_t)
        elif f == 5 and dtype == DT_FLOAT:
            chunks.append(np.frombuffer(v, dtype="<f4"))             # packed run (wire type 2) or one fixed32
        elif f == 6 and dtype == DT_DOUBLE:
            chunks.append(np.frombuffer(v, dtype="<f8"))
        elif f == 7 and dtype == DT_INT32:
            vals = _packed_varints(v, 64) if w == 2 else [v if v < (1 << 63) else v - (1 << 64)]
            chunks.append(np.array(vals, dtype=np.int64).astype(np.int32))
        elif f == 10 and dtype == DT_INT64:
            vals = _packed_varints(v, 64) if w == 2 else [v if v < (1 << 63) else v - (1 << 64)]
            chunks.append(np.array(vals, dtype=np.int64))
    if content is not None and content.size:
        return content
    return np.concatenate(chunks) if chunks else np.zeros(0, np_t)


def read_tf_v1_checkpoint(path, verify_checksums=False, names=None):
    """{variable name: ndarray} of every float/double/int32/int64 tensor in a TF V1 checkpoint file.
    `names`: optional predicate(name) -> bool to skip tensors."""
    with open(path, "rb") as f:
        data = f.read()
    meta, out, filled = {}, {}, {}
    for key, value in table_entries(data, verify_checksums):
        for f, w, v in _fields(value):
            if f == 1 and key == b"":                    # SavedTensorSliceMeta
                for f2, _, v2 in _fields(v):
                    if f2 != 1:
                        continue
                    name, shape, dtype = None, (), DT_FLOAT
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            name = v3.decode()
                        elif f3 == 2:
                            shape = _shape(v3)
                        elif f3 == 3:
                            dtype = v3
                    meta[name] = (shape, dtype)
            elif f == 2:                                 # SavedSlice
                name, sl, tp = None, b"", b""
                for f2, _, v2 in _fields(v):
                    if f2 == 1:
                        name = v2.decode()
                    elif f2 == 2:
                        sl = v2
                    elif f2 == 3:
                        tp = v2
                if name not in meta:
                    raise ValueError("slice of %r precedes / lacks its metadata entry" % name)
                shape, dtype = meta[name]
                if dtype not in _NP or (names is not None and not names(name)):
                    continue
                if name not in out:
                    out[name] = np.zeros(shape, dtype=_NP[dtype])
                    filled[name] = 0
                region = _extents(sl, shape)
                vals = _tensor_values(tp, dtype)
                target = out[name][region] if shape else out[name]
                if vals.size != target.size:
                    raise ValueError("%s: slice holds %d values, expected %d" % (name, vals.size, target.size))
                if shape:
                    out[name][region] = vals.reshape(target.shape)
                else:
                    out[name][...] = vals.reshape(())
                filled[name] += vals.size
    for name, n in filled.items():
        if n != out[name].size:
            raise ValueError("%s: %d of %d values present" % (name, n, out[name].size))
    return out


# ---- writer (tests, and exporting weights back to a TF-readable file) ------------------------------------------
def _shape_proto(shape):
    return b"".join(_len_field(2, _varint(1 << 3 | 0) + _varint(int(d))) for d in shape)


def _block(entries, restart_interval=16):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_tf_v1_checkpoint(path, tensors, block_bytes=1 << 16, max_slice_elems=None):
    """Write {name: ndarray} in the layout read above.  Data keys are b'\\x00' + name + slice index (a stand-in
    for TensorFlow's OrderedCode keys, which no reader needs).  `max_slice_elems` splits tensors along their
    first dimension into several slices (exercises slice assembly)."""
    names = sorted(tensors)
    metas, items = [], []
    for name in names:
        a = np.asarray(tensors[name])
        dtype = {np.dtype(np.float32): DT_FLOAT, np.dtype(np.float64): DT_DOUBLE, np.dtype(np.int32): DT_INT32,
                 np.dtype(np.int64): DT_INT64}[a.dtype]
        metas.append(_len_field(1, _len_field(1, name.encode()) + _len_field(2, _shape_proto(a.shape)) +
                                _varint(3 << 3 | 0) + _varint(dtype)))
        rows = a.shape[0] if a.ndim else 1
        step = rows if not max_slice_elems or a.ndim == 0 else max(1, max_slice_elems // max(1, a.size // max(rows, 1)))
        for si, r0 in enumerate(range(0, rows, step)):
            part = a[r0:r0 + step] if a.ndim else a
            ext = b""
            if a.ndim:
                ext = _len_field(1, _varint(1 << 3 | 0) + _varint(r0) + _varint(2 << 3 | 0) + _varint(part.shape[0]))
                ext += b"".join(_len_field(1, b"") for _ in range(a.ndim - 1))       # full extents
            flat = np.ascontiguousarray(part).reshape(-1)
            if dtype == DT_FLOAT:
                payload = _len_field(5, flat.astype("<f4").tobytes())
            elif dtype == DT_DOUBLE:
                payload = _len_field(6, flat.astype("<f8").tobytes())
            else:
                payload = _len_field(7 if dtype == DT_INT32 else 10,
                                     b"".join(_varint(int(x) & ((1 << 64) - 1)) for x in flat))
            tp = _varint(1 << 3 | 0) + _varint(dtype) + payload
            saved = _len_field(1, name.encode()) + _len_field(2, ext) + _len_field(3, tp)
            items.append((b"\x00" + name.encode() + b"\x00" + struct.pack(">I", si), _len_field(2, saved)))
    entries = [(b"", _len_field(1, b"".join(metas)))] + items
    out, index = bytearray(), []

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)                                        # no compression
        out.extend(struct.pack("<I", masked_crc(bytes(block) + b"\x00")))
        return _varint(off) + _varint(len(block))

    cur, size = [], 0
    for k, v in entries:
        cur.append((k, v))
        size += len(k) + len(v)
        if size >= block_bytes:
            index.append((cur[-1][0], emit(_block(cur))))
            cur, size = [], 0
    if cur:
        index.append((cur[-1][0], emit(_block(cur))))
    meta_h = emit(_block([]))
    index_h = emit(_block(index, restart_interval=1))
    footer = meta_h + index_h
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC))
    with open(path, "wb") as f:
        f.write(bytes(out))
