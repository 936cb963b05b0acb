"""TensorFlow checkpoints ("tensor bundles": ``<prefix>.index`` + ``<prefix>.data-00000-of-00001``) without TensorFlow
(SURVEY 8 f-1).

The reference's trainer starts from and writes this container: ``model.load_weights(f"{args.model_id}/tf_model")``
(``src/main.py:132``) and ``ModelCheckpoint(filepath=".../tf_model", save_weights_only=True)``
(``src/training_utils.py:32-45``) -- a Keras ``save_weights`` / ``load_weights`` on a path without an ``.h5`` suffix.
This module reads and writes it from the published formats:

* ``.index`` is a sorted string table in the LevelDB table format TensorFlow vendors (``tensorflow/core/lib/io/table*``,
  ``format.cc``, ``block_builder.cc``): data blocks of prefix-compressed entries
  ``varint32 shared | varint32 unshared | varint32 value_len | key tail | value`` with a restart array
  (``uint32`` offsets + count; interval 16), each block followed by a 5-byte trailer (compression type, masked CRC-32C of
  block + type), an (empty) metaindex block, an index block (interval 1) of ``separator key -> BlockHandle(varint64 offset,
  varint64 size)``, and a 48-byte footer (the two handles, zero padding to 40 bytes, magic ``0xdb4775248b80fb57``).
  TensorFlow writes the index uncompressed; a snappy block raises.
* key ``""`` holds ``BundleHeaderProto{num_shards=1, endianness=2, version=3: VersionDef{producer=1}}``; every other key is
  a tensor name with ``BundleEntryProto{dtype=1, shape=2: TensorShapeProto{dim=2: {size=1}}, shard_id=3, offset=4, size=5,
  crc32c=6 (fixed32, masked CRC-32C of the tensor's bytes)}`` (``tensorflow/core/protobuf/tensor_bundle.proto``);
* the data shard is the tensors' little-endian bytes back to back in key order.  A ``DT_STRING`` tensor is
  ``varint64 length per element | fixed32 masked CRC-32C of the lengths | the bytes``;
* an object-based (TF2 / Keras) checkpoint names its tensors by attribute paths
  (``layer_with_weights-0/.../kernel/.ATTRIBUTES/VARIABLE_VALUE``) and stores, under ``_CHECKPOINTABLE_OBJECT_GRAPH``, a
  serialized ``TrackableObjectGraph{nodes=1: TrackableObject{children=1, attributes=2: SerializedTensor{name=1,
  full_name=2, checkpoint_key=3}}}`` (``trackable_object_graph.proto``).  ``full_name`` is the variable's own name, so the
  reader maps **variable name -> checkpoint key** through it and never needs the Python object structure of the writer.

The writer emits the name-based form (keys = variable names, no object graph), which Keras' ``load_weights`` restores by
variable name (``NameBasedSaverStatus``); ``object_graph=True`` writes an object-based file whose object tree is the
variable names' own ``/`` hierarchy (used by the tests to exercise that read path).

PARITY NOTE -- unpinned: TensorFlow is not installable here and the reference ships no checkpoint file, so no bundle
written by TensorFlow itself was available.  The CRC is pinned by the RFC 3720 vectors, the byte layout of a one-entry
table by a hand-assembled known answer from the format definitions above (tests/test_host_cpu.py), everything else by
round trips.
"""

import os
import struct

import numpy as np

from .tfrecord import _TABLE, _fields, _ld, _read_varint, _varint

HEADER_KEY = b""
OBJECT_GRAPH_KEY = b"_CHECKPOINTABLE_OBJECT_GRAPH"
TABLE_MAGIC = 0xDB4775248B80FB57
BLOCK_SIZE = 262144            # table::Options::block_size
RESTART_INTERVAL = 16          # table::Options::block_restart_interval
_MASK_DELTA = 0xA282EAD8

DT_FLOAT, DT_DOUBLE, DT_INT32, DT_STRING, DT_INT64, DT_BOOL, DT_BFLOAT16, DT_HALF = 1, 2, 3, 7, 9, 10, 14, 19
_NP_OF = {DT_FLOAT: np.dtype("<f4"), DT_DOUBLE: np.dtype("<f8"), DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8"),
          DT_BOOL: np.dtype("bool"), DT_BFLOAT16: np.dtype("<u2"), DT_HALF: np.dtype("<f2")}
_DT_OF = {np.dtype("float32"): DT_FLOAT, np.dtype("float64"): DT_DOUBLE, np.dtype("int32"): DT_INT32,
          np.dtype("int64"): DT_INT64, np.dtype("bool"): DT_BOOL, np.dtype("float16"): DT_HALF}


# ---- CRC-32C: the native library's slicing-by-8 when it is there (a base model is 0.38 GB), else the byte loop ----------
_NATIVE_CRC = None


def _crc32c(data, crc=0) -> int:
    """CRC-32C of `data` continuing from `crc` (crc32c::Extend)."""
    global _NATIVE_CRC
    if _NATIVE_CRC is None:
        try:
            from . import _native
            _NATIVE_CRC = _native.load(build_if_missing=False).w2v2_crc32c_extend
        except Exception:       # no built library in this process (pure-host use): same function, a byte at a time
            _NATIVE_CRC = False
    if _NATIVE_CRC:
        buf = data.view(np.uint8).reshape(-1) if isinstance(data, np.ndarray) else np.frombuffer(bytes(data), dtype=np.uint8)
        buf = np.ascontiguousarray(buf)
        return int(_NATIVE_CRC(crc, buf.ctypes.data if buf.size else None, buf.size)) & 0xFFFFFFFF
    c = crc ^ 0xFFFFFFFF
    for b in bytes(data):
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xFFFFFFFF


# ---- the string table --------------------------------------------------------------------------------------------------
class _BlockBuilder:
    def __init__(self, interval):
        self.interval, self.buf, self.restarts, self.count, self.last = interval, bytearray(), [0], 0, b""

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.count < self.interval:
            n = min(len(self.last), len(key))
            while shared < n and self.last[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last, self.count = key, self.count + 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self):
        return not self.buf

    def finish(self) -> bytes:
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _shortest_separator(start: bytes, limit: bytes) -> bytes:
    """BytewiseComparator::FindShortestSeparator: a short key k with start <= k < limit."""
    n = min(len(start), len(limit))
    d = 0
    while d < n and start[d] == limit[d]:
        d += 1
    if d < n and start[d] < 0xFF and start[d] + 1 < limit[d]:
        return start[:d] + bytes([start[d] + 1])
    return start


def _short_successor(key: bytes) -> bytes:
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


def _handle(offset, size) -> bytes:
    return _varint(offset) + _varint(size)


def write_table(path, items, block_size=BLOCK_SIZE, restart_interval=RESTART_INTERVAL):
    """items: iterable of (key bytes, value bytes) in strictly increasing key order."""
    out = bytearray()
    index = _BlockBuilder(1)
    data = _BlockBuilder(restart_interval)
    pending = None                  # (last key of the flushed block, its handle)
    last_key = None

    def emit(block: bytes):
        off = len(out)
        out.extend(block)
        out.extend(b"\x00" + struct.pack("<I", _mask(_crc32c(b"\x00", _crc32c(block)))))       # no compression
        return off, len(block)

    for key, value in items:
        if last_key is not None and key <= last_key:
            raise ValueError("write_table: keys must be strictly increasing")
        if pending is not None:
            index.add(_shortest_separator(pending[0], key), _handle(*pending[1]))
            pending = None
        data.add(key, value)
        last_key = key
        if data.size() >= block_size:
            pending = (last_key, emit(data.finish()))
            data = _BlockBuilder(restart_interval)
    if not data.empty():
        pending = (last_key, emit(data.finish()))
    meta = emit(_BlockBuilder(restart_interval).finish())
    if pending is not None:
        index.add(_short_successor(pending[0]), _handle(*pending[1]))
    idx = emit(index.finish())
    footer = _handle(*meta) + _handle(*idx)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<II", TABLE_MAGIC & 0xFFFFFFFF, TABLE_MAGIC >> 32)
    out.extend(footer)
    with open(path, "wb") as f:
        f.write(out)


def _read_block(buf, offset, size, check_crc):
    if offset + size + 5 > len(buf):
        raise ValueError("table: block handle points outside the file")
    block, kind = buf[offset:offset + size], buf[offset + size]
    if check_crc:
        want = struct.unpack_from("<I", buf, offset + size + 1)[0]
        if _mask(_crc32c(bytes([kind]), _crc32c(block))) != want:
            raise ValueError("table: block checksum mismatch")
    if kind != 0:
        raise ValueError("table: compressed block (type %d); TensorFlow writes checkpoint indexes uncompressed" % kind)
    return block


def _block_entries(block):
    if len(block) < 4:
        raise ValueError("table: block too short")
    nrestarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestarts
    if end < 0:
        raise ValueError("table: bad restart count")
    pos, key = 0, b""
    while pos < end:
        shared, pos = _read_varint(block, pos)
        unshared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        if shared > len(key) or pos + unshared + vlen > end:
            raise ValueError("table: corrupt entry")
        key = key[:shared] + bytes(block[pos:pos + unshared])
        pos += unshared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, check_crc=True):
    """{key: value} of a table file, in key order."""
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < 48:
        raise ValueError(f"{path}: too short for a table")
    lo, hi = struct.unpack_from("<II", buf, len(buf) - 8)
    if (hi << 32 | lo) != TABLE_MAGIC:
        raise ValueError(f"{path}: not a TensorFlow / LevelDB table (bad magic)")
    foot = buf[len(buf) - 48:len(buf) - 8]
    pos = 0
    _, pos = _read_varint(foot, pos)            # metaindex handle (no filter / properties in checkpoint indexes)
    _, pos = _read_varint(foot, pos)
    ioff, pos = _read_varint(foot, pos)
    isize, pos = _read_varint(foot, pos)
    out = {}
    for _, handle in _block_entries(_read_block(buf, ioff, isize, check_crc)):
        off, p = _read_varint(handle, 0)
        size, p = _read_varint(handle, p)
        for key, value in _block_entries(_read_block(buf, off, size, check_crc)):
            out[key] = value
    return out


# ---- bundle protos ------------------------------------------------------------------------------------------------------
def _header_proto() -> bytes:
    return _varint(1 << 3) + _varint(1) + _ld(3, _varint(1 << 3) + _varint(1))       # num_shards = 1, version.producer = 1


def _entry_proto(dtype, shape, offset, size, crc) -> bytes:
    dims = b"".join(_ld(2, (_varint(1 << 3) + _varint(int(d))) if d else b"") for d in shape)
    out = _varint(1 << 3) + _varint(dtype) + _ld(2, dims)
    if offset:
        out += _varint(4 << 3) + _varint(offset)
    if size:
        out += _varint(5 << 3) + _varint(size)
    if crc:                                             # (proto3: zero-valued scalars are not serialized)
        out += _varint((6 << 3) | 5) + struct.pack("<I", crc)
    return out


def _parse_entry(buf):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=0, sliced=False)
    for field, wt, val in _fields(buf):
        if field == 1 and wt == 0:
            e["dtype"] = val
        elif field == 2 and wt == 2:
            for f2, w2, dim in _fields(val):
                if f2 == 2 and w2 == 2:
                    size = 0
                    for f3, w3, v3 in _fields(dim):
                        if f3 == 1 and w3 == 0:
                            size = v3
                    e["shape"].append(size)
        elif field == 3 and wt == 0:
            e["shard_id"] = val
        elif field == 4 and wt == 0:
            e["offset"] = val
        elif field == 5 and wt == 0:
            e["size"] = val
        elif field == 6 and wt == 5:
            e["crc32c"] = struct.unpack("<I", val)[0]
        elif field == 7:
            e["sliced"] = True
    return e


def _parse_header(buf):
    h = dict(num_shards=0, endianness=0, producer=0)
    for field, wt, val in _fields(buf):
        if field == 1 and wt == 0:
            h["num_shards"] = val
        elif field == 2 and wt == 0:
            h["endianness"] = val
        elif field == 3 and wt == 2:
            for f2, w2, v2 in _fields(val):
                if f2 == 1 and w2 == 0:
                    h["producer"] = v2
    return h


def _object_graph_names(graph: bytes):
    """{variable full_name: checkpoint_key} over every SerializedTensor of a TrackableObjectGraph."""
    out = {}
    for field, wt, node in _fields(graph):
        if field != 1 or wt != 2:
            continue
        for f2, w2, attr in _fields(node):
            if f2 != 2 or w2 != 2:
                continue
            name = full = key = ""
            for f3, w3, v3 in _fields(attr):
                if w3 != 2:
                    continue
                if f3 == 1:
                    name = v3.decode()
                elif f3 == 2:
                    full = v3.decode()
                elif f3 == 3:
                    key = v3.decode()
            if name == "VARIABLE_VALUE" and full and key:
                out.setdefault(full, key)
    return out


def _shard_name(prefix, shard, num_shards):
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


class BundleReader:
    """Random access to the tensors of a checkpoint by CHECKPOINT KEY (`keys()`, `tensor(key)`), and the variable-name view
    object-based checkpoints carry (`variables()`)."""

    def __init__(self, prefix, check_crc=True):
        self.prefix, self.check_crc = prefix, check_crc
        if not os.path.exists(prefix + ".index"):
            raise FileNotFoundError(f"{prefix}.index: no such TensorFlow checkpoint")
        table = read_table(prefix + ".index", check_crc)
        if HEADER_KEY not in table:
            raise ValueError(f"{prefix}.index: no bundle header")
        self.header = _parse_header(table.pop(HEADER_KEY))
        if self.header["endianness"] != 0:
            raise ValueError("big-endian checkpoints are not supported")
        self.entries = {k.decode(): _parse_entry(v) for k, v in table.items()}
        self._shards = {}

    def keys(self):
        return list(self.entries)

    def _shard(self, i):
        if i not in self._shards:
            self._shards[i] = np.memmap(_shard_name(self.prefix, i, self.header["num_shards"]), dtype=np.uint8, mode="r")
        return self._shards[i]

    def _bytes(self, e):
        data = self._shard(e["shard_id"])
        if e["offset"] + e["size"] > data.size:
            raise ValueError("checkpoint data shard is shorter than its index says")
        return data[e["offset"]:e["offset"] + e["size"]]

    def tensor(self, key):
        e = self.entries[key]
        if e["sliced"]:
            raise ValueError(f"{key}: partitioned (sliced) variables are not supported")
        raw = self._bytes(e)
        if e["dtype"] == DT_STRING:
            n = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
            buf, pos, lens = bytes(raw), 0, []
            for _ in range(n):
                ln, pos = _read_varint(buf, pos)
                lens.append(ln)
            pos += 4                                        # masked CRC-32C of the lengths
            vals = []
            for ln in lens:
                vals.append(buf[pos:pos + ln])
                pos += ln
            return np.array(vals, dtype=object).reshape(e["shape"])
        if e["dtype"] not in _NP_OF:
            raise ValueError(f"{key}: unsupported dtype enum {e['dtype']}")
        if self.check_crc and _mask(_crc32c(raw)) != e["crc32c"]:
            raise ValueError(f"{key}: tensor checksum mismatch")
        a = np.frombuffer(raw, dtype=_NP_OF[e["dtype"]])
        want = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if a.size != want:
            raise ValueError(f"{key}: {a.size} elements on disk, shape {e['shape']}")
        a = a.reshape(e["shape"])
        if e["dtype"] == DT_BFLOAT16:
            a = (a.astype(np.uint32) << 16).view(np.float32)
        return np.array(a)

    def variables(self):
        """{variable name: checkpoint key}: through the object graph's `full_name`s when there is one, else the keys themselves
        (a name-based checkpoint)."""
        if OBJECT_GRAPH_KEY.decode() in self.entries:
            graph = self.tensor(OBJECT_GRAPH_KEY.decode()).reshape(-1)[0]
            return _object_graph_names(graph)
        return {k: k for k in self.entries}


def read_checkpoint(prefix, check_crc=True):
    """{variable name: array} of every numeric variable of the checkpoint."""
    r = BundleReader(prefix, check_crc)
    out = {}
    for name, key in r.variables().items():
        if key in r.entries and r.entries[key]["dtype"] != DT_STRING:
            out[name] = r.tensor(key)
    return out


def _object_graph_for(names):
    """A TrackableObjectGraph whose tree is the `/` hierarchy of the variable names; returns (proto bytes, {name: key})."""
    nodes = [dict(children=[], attrs=[])]
    index = {(): 0}
    keys = {}
    for name in names:
        parts = tuple(name.split("/"))
        for d in range(1, len(parts) + 1):
            if parts[:d] not in index:
                index[parts[:d]] = len(nodes)
                nodes.append(dict(children=[], attrs=[]))
                nodes[index[parts[:d - 1]]]["children"].append((index[parts[:d]], parts[d - 1]))
        key = name + "/.ATTRIBUTES/VARIABLE_VALUE"
        keys[name] = key
        nodes[index[parts]]["attrs"].append(("VARIABLE_VALUE", name, key))
    out = b""
    for n in nodes:
        body = b"".join(_ld(1, (_varint(1 << 3) + _varint(i) if i else b"") + _ld(2, ln.encode())) for i, ln in n["children"])
        body += b"".join(_ld(2, _ld(1, a.encode()) + _ld(2, f.encode()) + _ld(3, k.encode())) for a, f, k in n["attrs"])
        out += _ld(1, body)
    return out, keys


def write_checkpoint(prefix, tensors, object_graph=False):
    """One-shard bundle of {variable name: array}.  Name-based by default (keys = names); `object_graph=True` writes
    `<name>/.ATTRIBUTES/VARIABLE_VALUE` keys plus the `_CHECKPOINTABLE_OBJECT_GRAPH` string tensor."""
    items = {}
    graph = None
    if object_graph:
        graph, keymap = _object_graph_for(sorted(tensors))
        for name, a in tensors.items():
            items[keymap[name]] = a
    else:
        items = dict(tensors)
    order = sorted(list(items) + ([OBJECT_GRAPH_KEY.decode()] if graph is not None else []), key=lambda k: k.encode())
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    entries = [(HEADER_KEY, _header_proto())]
    offset = 0
    with open(_shard_name(prefix, 0, 1), "wb") as f:
        for key in order:
            if graph is not None and key == OBJECT_GRAPH_KEY.decode():
                lengths = _varint(len(graph))
                c = _crc32c(struct.pack("<I", len(graph)))                     # lengths enter the checksum as uint32
                lcrc = struct.pack("<I", _mask(c))
                c = _crc32c(graph, _crc32c(lcrc, c))
                raw = lengths + lcrc + graph
                f.write(raw)
                entries.append((key.encode(), _entry_proto(DT_STRING, [], offset, len(raw), _mask(c))))
                offset += len(raw)
                continue
            a = np.asarray(items[key])
            if a.dtype not in _DT_OF:
                raise TypeError(f"{key}: dtype {a.dtype} has no TensorFlow checkpoint encoding here")
            shape = a.shape                                 # (ascontiguousarray would turn a scalar into shape (1,))
            raw = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<"), copy=False)).reshape(-1).view(np.uint8)
            f.write(raw.tobytes())
            entries.append((key.encode(), _entry_proto(_DT_OF[a.dtype], shape, offset, raw.size, _mask(_crc32c(raw)))))
            offset += raw.size
    write_table(prefix + ".index", entries)


def is_checkpoint(prefix) -> bool:
    return os.path.exists(prefix + ".index")
