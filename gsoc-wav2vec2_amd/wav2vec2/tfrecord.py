"""TFRecord files of the reference's schema, without TensorFlow.

The reference stores its training sets as TFRecords (``src/make_tfrecords.py:10-23``) and reads them back in
``src/data_utils.py:17-27``: one ``tf.train.Example`` per utterance with two ``bytes`` features, ``"speech"`` and
``"label"``, each holding ``tf.io.serialize_tensor`` of a float32 waveform / an int32 label vector.  This module
reads and writes exactly that, from the published formats:

* TFRecord framing: ``uint64 length | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data)``, little
  endian, CRC-32C (Castagnoli) masked as ``rotr15(crc) + 0xa282ead8``;
* ``Example{features=1: Features{feature=1: map<string, Feature{bytes_list=1: BytesList{value=1}}>}}``;
* ``serialize_tensor`` = a serialized ``TensorProto{dtype=1, tensor_shape=2: {dim=2: {size=1}}, tensor_content=4}``
  with ``DT_FLOAT = 1``, ``DT_INT32 = 3`` and the elements little-endian in ``tensor_content``.

Only the protobuf wire subset those three messages use is implemented (varint and length-delimited fields); unknown
fields are skipped, as a protobuf parser would.  PARITY NOTE: TensorFlow is not installable here, so no file written
by the reference itself was available; the CRC is pinned by the RFC 3720 test vectors and the layout by the format
definitions above (tests/test_host_cpu.py).
"""

import struct

import numpy as np

DT_FLOAT, DT_INT32, DT_INT64 = 1, 3, 9
_DTYPES = {DT_FLOAT: np.dtype("<f4"), DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8")}
_DT_OF = {np.dtype("float32"): DT_FLOAT, np.dtype("int32"): DT_INT32, np.dtype("int64"): DT_INT64}


# ---- CRC-32C --------------------------------------------------------------------------------------------------
def _make_table():
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t[i] = c
    return t


_TABLE = [int(v) for v in _make_table()]


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protobuf wire subset -------------------------------------------------------------------------------------
def _varint(n: int) -> bytes:
    if n < 0:
        n += 1 << 64
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _read_varint(buf, pos):
    shift = val = 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _ld(field: int, payload: bytes) -> bytes:          # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _fields(buf):
    """Yield (field number, wire type, value) over one message; value is an int (varint) or bytes."""
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _read_varint(buf, pos)
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            if pos + n > len(buf):
                raise ValueError("truncated length-delimited field")
            val, pos = bytes(buf[pos:pos + n]), pos + n
        elif wt == 1:
            val, pos = bytes(buf[pos:pos + 8]), pos + 8
        elif wt == 5:
            val, pos = bytes(buf[pos:pos + 4]), pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield field, wt, val


# ---- tf.io.serialize_tensor / parse_tensor --------------------------------------------------------------------
def serialize_tensor(a) -> bytes:
    a = np.asarray(a)
    if a.dtype not in _DT_OF:
        raise TypeError(f"serialize_tensor: unsupported dtype {a.dtype}")
    shape = b"".join(_ld(2, _varint((1 << 3) | 0) + _varint(int(d))) for d in a.shape)
    content = np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
    return _varint((1 << 3) | 0) + _varint(_DT_OF[a.dtype]) + _ld(2, shape) + _ld(4, content)


def parse_tensor(buf: bytes, out_type=None) -> np.ndarray:
    """``tf.io.parse_tensor``: raises ``ValueError`` when the stored dtype is not ``out_type`` (TF raises
    InvalidArgumentError there)."""
    dtype, dims, content = None, [], b""
    for field, wt, val in _fields(buf):
        if field == 1 and wt == 0:
            dtype = val
        elif field == 2 and wt == 2:
            for f2, w2, v2 in _fields(val):
                if f2 == 2 and w2 == 2:
                    size = 0
                    for f3, w3, v3 in _fields(v2):
                        if f3 == 1 and w3 == 0:
                            size = v3 - (1 << 64) if v3 >> 63 else v3
                    dims.append(size)
        elif field == 4 and wt == 2:
            content = val
    if dtype not in _DTYPES:
        raise ValueError(f"parse_tensor: unsupported or missing dtype {dtype}")
    np_dtype = _DTYPES[dtype]
    if out_type is not None and np.dtype(out_type) != np.dtype(np_dtype.name):
        raise ValueError(f"parse_tensor: stored dtype {np_dtype.name} does not match out_type {np.dtype(out_type).name}")
    n = int(np.prod(dims)) if dims else 1
    if len(content) != n * np_dtype.itemsize:
        raise ValueError("parse_tensor: tensor_content does not match the shape (typed value fields are not supported)")
    return np.frombuffer(content, dtype=np_dtype).reshape(dims).astype(np_dtype.name)


# ---- tf.train.Example with bytes features ---------------------------------------------------------------------
def encode_example(features: dict) -> bytes:
    """{name: bytes} -> serialized Example (each a one-element BytesList, as make_tfrecords.py:17-23 builds them)."""
    entries = b""
    for name in sorted(features):                       # protobuf serializes map entries in key order deterministically
        feat = _ld(1, _ld(1, bytes(features[name])))    # Feature{bytes_list = BytesList{value = [...]}}
        entries += _ld(1, _ld(1, name.encode()) + _ld(2, feat))
    return _ld(1, entries)


def decode_example(buf: bytes) -> dict:
    out = {}
    for f, wt, features in _fields(buf):
        if f != 1 or wt != 2:
            continue
        for f1, w1, entry in _fields(features):
            if f1 != 1 or w1 != 2:
                continue
            key, values = None, []
            for f2, w2, v2 in _fields(entry):
                if f2 == 1 and w2 == 2:
                    key = v2.decode()
                elif f2 == 2 and w2 == 2:
                    for f3, w3, v3 in _fields(v2):
                        if f3 == 1 and w3 == 2:                   # bytes_list
                            values = [v4 for f4, w4, v4 in _fields(v3) if f4 == 1 and w4 == 2]
            if key is not None:
                out[key] = values
    return out


# ---- files ----------------------------------------------------------------------------------------------------
def write_records(path, payloads):
    with open(path, "wb") as f:
        for data in payloads:
            head = struct.pack("<Q", len(data))
            f.write(head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data)))


def read_records(path, check_crc=True):
    with open(path, "rb") as f:
        while True:
            head = f.read(8)
            if not head:
                return
            if len(head) < 8:
                raise ValueError(f"{path}: truncated record header")
            (n,) = struct.unpack("<Q", head)
            crc_h = f.read(4)
            data = f.read(n)
            crc_d = f.read(4)
            if len(crc_h) < 4 or len(data) < n or len(crc_d) < 4:
                raise ValueError(f"{path}: truncated record")
            if check_crc and (struct.unpack("<I", crc_h)[0] != masked_crc32c(head) or struct.unpack("<I", crc_d)[0] != masked_crc32c(data)):
                raise ValueError(f"{path}: corrupted record (CRC mismatch)")
            yield data


def create_tfrecord(speech, label) -> bytes:
    """One utterance -> serialized Example (make_tfrecords.py:10-23): speech cast to float32, label to int32."""
    return encode_example({"speech": serialize_tensor(np.asarray(speech, np.float32)),
                           "label": serialize_tensor(np.asarray(label, np.int32))})


def read_tfrecords(record: bytes):
    """Serialized Example -> (speech float32, label int32) (data_utils.py:17-27)."""
    ex = decode_example(record)
    for key in ("speech", "label"):
        if len(ex.get(key, [])) != 1:
            raise ValueError(f"read_tfrecords: feature `{key}` missing or not a single bytes value")
    return parse_tensor(ex["speech"][0], np.float32), parse_tensor(ex["label"][0], np.int32)


def write_dataset(path, samples):
    """Iterable of (speech, label) -> one .tfrecord shard."""
    write_records(path, (create_tfrecord(s, l) for s, l in samples))


def read_dataset(paths, check_crc=True):
    """Yield (speech, label) over one or more shards, in file order (the reference interleaves and shuffles with
    tf.data on top; that is harness, not format)."""
    if isinstance(paths, (str, bytes)):
        paths = [paths]
    for p in paths:
        for rec in read_records(p, check_crc=check_crc):
            yield read_tfrecords(rec)
