"""A dependency-free reader / writer for the subset of HDF5 that Keras' ``save_weights("*.h5")`` uses
(reference ``src/wav2vec2/modeling.py:22-27,76-84``: ``tf_model.h5``).

This image has neither h5py nor TensorFlow, so the container format itself is implemented here from the
HDF5 File Format Specification (version 0 superblock, "earliest" library format -- what h5py / Keras write by
default):

    superblock v0 -> root symbol-table entry -> object header v1
    groups   = Symbol Table message -> v1 B-tree ("TREE") of symbol-table nodes ("SNOD") + local heap ("HEAP")
    datasets = Dataspace + Datatype + Fill Value + Layout(v3 contiguous | compact) messages
    attributes = Attribute messages (v1 written; v1 / v2 / v3 read) with fixed-length string / numeric arrays
    object-header continuation blocks are followed when reading

Pinned against the real library in the build container (libhdf5 1.10.6 under /opt/conda, driven through ctypes by
tests/golden/make_h5_fixture.py): a file written BY libhdf5 in the Keras layout is committed as a fixture and read
by this module bit-exactly; files written by this module were listed and dumped with the library's own ``h5ls`` /
``h5dump`` (tests/test_h5lite_cpu.py holds the structural checks that run anywhere).  Not supported (raises):
chunked / filtered datasets, new-style (link-message / fractal-heap) groups, variable-length strings, superblock
versions 2 and 3 -- none of which Keras' HDF5 weight files contain.

Keras layout on top (``keras/saving/hdf5_format.py`` as of TF 2.5, restated from its documented behaviour):
root attributes ``layer_names`` (fixed-length byte strings), ``backend``, ``keras_version``; one group per layer
with the attribute ``weight_names`` (chunked into ``weight_names0..n`` when it would exceed the 64 KiB object-header
limit) and one dataset per weight at ``<layer>/<weight name>`` -- weight names contain ``/``, so they nest as groups.
"""

import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 4, 16                      # superblock defaults: SNOD holds 2 * 4 entries, a B-tree node 2 * 16 children
HEADER_LIMIT = 64512                             # Keras' HDF5_OBJECT_HEADER_LIMIT

MSG_NIL, MSG_DATASPACE, MSG_LINK_INFO, MSG_DATATYPE, MSG_FILL_OLD, MSG_FILL, MSG_LINK = 0x0, 0x1, 0x2, 0x3, 0x4, 0x5, 0x6
MSG_LAYOUT, MSG_GROUP_INFO, MSG_FILTER, MSG_ATTRIBUTE, MSG_CONTINUATION, MSG_SYMBOL_TABLE = 0x8, 0xA, 0xB, 0xC, 0x10, 0x11
MSG_ATTR_INFO = 0x15


class H5FormatError(ValueError):
    pass


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


# ---------------------------------------------------------------------------------------------------------
# datatype / dataspace messages
# ---------------------------------------------------------------------------------------------------------
def _encode_datatype(dtype):
    dtype = np.dtype(dtype)
    if dtype.kind == "f" and dtype.itemsize in (4, 8):
        if dtype.itemsize == 4:
            bits, props = (0x20, 0x1F, 0x00), struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        else:
            bits, props = (0x20, 0x3F, 0x00), struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        return bytes([0x11, *bits]) + struct.pack("<I", dtype.itemsize) + props
    if dtype.kind in "iu":
        bits = (0x08 if dtype.kind == "i" else 0x00, 0, 0)
        return bytes([0x10, *bits]) + struct.pack("<I", dtype.itemsize) + struct.pack("<HH", 0, 8 * dtype.itemsize)
    if dtype.kind == "S":
        return bytes([0x13, 0x01, 0, 0]) + struct.pack("<I", max(1, dtype.itemsize))      # null-padded, ASCII
    raise TypeError(f"h5lite cannot store dtype {dtype}")


def _decode_datatype(buf, off):
    cls_ver, b0, b1, b2, size = struct.unpack_from("<BBBBI", buf, off)
    cls = cls_ver & 0x0F
    if cls == 1:                                   # floating point
        if b0 & 1:
            raise H5FormatError("big-endian floats are not supported")
        return np.dtype("<f%d" % size)
    if cls == 0:                                   # fixed point
        order = ">" if b0 & 1 else "<"
        return np.dtype("%s%s%d" % (order, "i" if b0 & 0x08 else "u", size))
    if cls == 3:                                   # fixed-length string
        return np.dtype("S%d" % size)
    if cls == 9:
        raise H5FormatError("variable-length datatypes are not supported")
    raise H5FormatError(f"datatype class {cls} is not supported")


def _encode_dataspace(shape):
    shape = tuple(int(d) for d in shape)
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", d) for d in shape)


def _decode_dataspace(buf, off):
    version = buf[off]
    if version == 1:
        rank, flags = buf[off + 1], buf[off + 2]
        p = off + 8
    elif version == 2:
        rank, flags, kind = buf[off + 1], buf[off + 2], buf[off + 3]
        if kind == 2:                              # null dataspace
            return (0,)
        p = off + 4
    else:
        raise H5FormatError(f"dataspace message version {version}")
    return tuple(struct.unpack_from("<Q", buf, p + 8 * i)[0] for i in range(rank))


# ---------------------------------------------------------------------------------------------------------
# writer
# ---------------------------------------------------------------------------------------------------------
class _Node:
    """In-memory tree handed to the writer: a group (children, attrs) or a dataset (array, attrs)."""

    def __init__(self, array=None):
        self.array = array
        self.children = {}
        self.attrs = {}

    def child_group(self, name):
        node = self.children.get(name)
        if node is None:
            node = self.children[name] = _Node()
        if node.array is not None:
            raise ValueError(f"`{name}` is a dataset, not a group")
        return node


class Writer:
    """Build a tree with ``create_group`` / ``create_dataset`` / ``attrs`` and write it with ``close()``.

        w = Writer(path)
        g = w.root.child_group("layer")
        w.set_attr(g, "weight_names", np.array([b"layer/kernel:0"]))
        w.create_dataset("layer/layer/kernel:0", array)
        w.close()
    """

    def __init__(self, path):
        self.path = path
        self.root = _Node()

    def group(self, path):
        node = self.root
        for part in [p for p in path.split("/") if p]:
            node = node.child_group(part)
        return node

    def create_dataset(self, path, array):
        parts = [p for p in path.split("/") if p]
        parent = self.group("/".join(parts[:-1]))
        if parts[-1] in parent.children:
            raise ValueError(f"`{path}` exists")
        node = parent.children[parts[-1]] = _Node(np.ascontiguousarray(array))
        return node

    @staticmethod
    def set_attr(node, name, value):
        node.attrs[name] = value

    # -- serialisation ------------------------------------------------------------------------------
    def close(self):
        with open(self.path, "wb") as f:
            self._f = f
            self._pos = 0
            self._emit(b"\0" * 96)                                     # superblock, patched at the end
            root_header, btree, heap = self._write_group(self.root)
            eof = self._pos
            sb = SIGNATURE + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", LEAF_K, INTERNAL_K, 0)
            sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
            sb += struct.pack("<QQII", 0, root_header, 1, 0) + struct.pack("<QQ", btree, heap)
            assert len(sb) == 96
            f.seek(0)
            f.write(sb)
        self._f = None

    def _emit(self, data):
        pad = -self._pos % 8
        if pad:
            self._f.write(b"\0" * pad)
            self._pos += pad
        addr = self._pos
        self._f.write(data)
        self._pos += len(data)
        return addr

    @staticmethod
    def _message(mtype, body, flags=0):
        body = _pad8(body)
        if len(body) > 0xFFFF:
            raise H5FormatError(f"object-header message of {len(body)} bytes exceeds the 64 KiB limit")
        return struct.pack("<HHB3x", mtype, len(body), flags) + body

    def _attribute_messages(self, attrs):
        out = []
        for name, value in attrs.items():
            arr = np.asarray(value)
            if arr.dtype.kind == "U":
                arr = np.char.encode(arr, "utf-8")
            if arr.dtype.kind == "S" and arr.dtype.itemsize == 0:
                arr = arr.astype("S1")
            if arr.dtype.kind == "O":
                raise TypeError(f"attribute `{name}`: object arrays are not supported")
            dt = _encode_datatype(arr.dtype)
            ds = _encode_dataspace(arr.shape)
            nm = name.encode("utf-8") + b"\0"
            body = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds)
            body += np.ascontiguousarray(arr).tobytes()
            out.append(self._message(MSG_ATTRIBUTE, body))
        return out

    def _object_header(self, messages):
        body = b"".join(messages)
        return self._emit(struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body)

    def _write_dataset(self, node):
        arr = node.array
        raw = arr.tobytes()
        addr = self._emit(raw) if raw else UNDEF
        msgs = [self._message(MSG_DATASPACE, _encode_dataspace(arr.shape)),
                self._message(MSG_DATATYPE, _encode_datatype(arr.dtype), flags=1),
                self._message(MSG_FILL, struct.pack("<BBBBI", 2, 2, 2, 1, 0), flags=1),
                self._message(MSG_LAYOUT, struct.pack("<BBQQ", 3, 1, addr, len(raw)))]
        msgs += self._attribute_messages(node.attrs)
        return self._object_header(msgs)

    def _write_group(self, node):
        # children first: their object-header addresses go into this group's symbol-table nodes
        names = sorted(node.children, key=lambda s: s.encode("utf-8"))
        entries = []
        for name in names:
            child = node.children[name]
            if child.array is not None:
                entries.append((name, self._write_dataset(child), 0, 0, 0))
            else:
                header, btree, heap = self._write_group(child)
                entries.append((name, header, 1, btree, heap))
        # local heap: the empty string at offset 0, then the member names, each padded to 8 bytes
        heap_data = bytearray(b"\0" * 8)
        offsets = {}
        for name in names:
            offsets[name] = len(heap_data)
            heap_data += _pad8(name.encode("utf-8") + b"\0")
        data_addr = self._emit(bytes(heap_data))
        heap_addr = self._emit(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 1, data_addr))     # 1 = no free block
        # symbol-table nodes of up to 2 * LEAF_K entries
        per = 2 * LEAF_K
        chunks = [entries[i:i + per] for i in range(0, len(entries), per)] or [[]]
        snods = []
        for chunk in chunks:
            body = b"SNOD" + struct.pack("<BBH", 1, 0, len(chunk))
            for name, header, cache, btree, heap in chunk:
                body += struct.pack("<QQII", offsets[name], header, cache, 0)
                body += struct.pack("<QQ", btree, heap) if cache == 1 else b"\0" * 16
            body += b"\0" * (40 * (per - len(chunk)))
            snods.append((self._emit(body), offsets[chunk[-1][0]] if chunk else 0))
        # v1 B-tree over the symbol-table nodes: up to 2 * INTERNAL_K children per node, key 0 = "" (heap offset 0),
        # key i+1 = the largest name below child i; more than 32 symbol-table nodes stack further levels on top
        fan = 2 * INTERNAL_K
        node_bytes = 24 + 8 * (2 * fan + 1)
        level, children = 0, snods                                     # children: [(address, heap offset of the largest name)]
        while True:
            groups = [children[i:i + fan] for i in range(0, len(children), fan)]
            pad = -self._pos % 8
            base = self._pos + pad
            addrs = [base + node_bytes * i for i in range(len(groups))]
            parents = []
            for i, grp in enumerate(groups):
                left = addrs[i - 1] if i > 0 else UNDEF
                right = addrs[i + 1] if i + 1 < len(groups) else UNDEF
                first_key = 0 if i == 0 else groups[i - 1][-1][1]
                tree = b"TREE" + struct.pack("<BBHQQ", 0, level, len(grp), left, right) + struct.pack("<Q", first_key)
                for addr, last_key in grp:
                    tree += struct.pack("<QQ", addr, last_key)
                tree += b"\0" * (16 * (fan - len(grp)))
                assert len(tree) == node_bytes
                got = self._emit(tree)
                assert got == addrs[i]
                parents.append((got, grp[-1][1]))
            if len(parents) == 1:
                btree_addr = parents[0][0]
                break
            level, children = level + 1, parents
        msgs = [self._message(MSG_SYMBOL_TABLE, struct.pack("<QQ", btree_addr, heap_addr))]
        msgs += self._attribute_messages(node.attrs)
        return self._object_header(msgs), btree_addr, heap_addr


# ---------------------------------------------------------------------------------------------------------
# reader
# ---------------------------------------------------------------------------------------------------------
class Object:
    """A group or dataset of an open file: ``.attrs`` (dict), ``.is_dataset``, ``.members()`` / ``[name]`` for groups,
    ``.read()`` for datasets."""

    def __init__(self, file, address):
        self._file = file
        self.address = address
        self.attrs = {}
        self._stab = None
        self._shape = self._dtype = self._layout = None
        file._parse_header(self)

    @property
    def is_dataset(self):
        return self._layout is not None

    def members(self):
        if self._stab is None:
            if self.is_dataset:
                raise H5FormatError("a dataset has no members")
            return {}
        return self._file._group_members(*self._stab)

    def __getitem__(self, path):
        obj = self
        for part in [p for p in path.split("/") if p]:
            members = obj.members()
            if part not in members:
                raise KeyError(path)
            obj = Object(self._file, members[part])
        return obj

    def read(self):
        if not self.is_dataset:
            raise H5FormatError("not a dataset")
        kind, a, b = self._layout
        count = int(np.prod(self._shape)) if self._shape else 1
        nbytes = count * self._dtype.itemsize
        if kind == "contiguous":
            if nbytes and (a == UNDEF or a + nbytes > len(self._file.buf)):
                raise H5FormatError("dataset storage lies outside the file")
            raw = self._file.buf[a:a + nbytes] if nbytes else b""
        else:
            raw = self._file.buf[a:a + nbytes]
        return np.frombuffer(raw, dtype=self._dtype, count=count).reshape(self._shape).copy()

    def visit_datasets(self, prefix=""):
        """{path: Object} of every dataset below this group."""
        out = {}
        for name, addr in self.members().items():
            child = Object(self._file, addr)
            path = prefix + name
            if child.is_dataset:
                out[path] = child
            else:
                out.update(child.visit_datasets(path + "/"))
        return out


class File:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.buf = f.read()
        if self.buf[:8] != SIGNATURE:
            raise H5FormatError(f"{path}: not an HDF5 file (bad signature)")
        version = self.buf[8]
        if version not in (0, 1):
            raise H5FormatError(f"{path}: superblock version {version} is not supported (Keras / h5py write version 0)")
        if self.buf[13] != 8 or self.buf[14] != 8:
            raise H5FormatError("only 8-byte offsets and lengths are supported")
        p = 24 if version == 0 else 28             # v1 adds the indexed-storage K and a reserved field
        base, _, self.eof, _ = struct.unpack_from("<QQQQ", self.buf, p)
        if base != 0:
            raise H5FormatError("a non-zero base address is not supported")
        _, root_header, _, _ = struct.unpack_from("<QQII", self.buf, p + 32)
        self.root = Object(self, root_header)

    def __getitem__(self, path):
        return self.root[path]

    # -- object headers -----------------------------------------------------------------------------
    def _parse_header(self, obj):
        buf = self.buf
        version = buf[obj.address]
        if version != 1:
            if buf[obj.address:obj.address + 4] == b"OHDR":
                raise H5FormatError("version-2 object headers (libver='latest') are not supported")
            raise H5FormatError(f"object header version {version}")
        nmsgs, _, size = struct.unpack_from("<HII", buf, obj.address + 2)
        blocks = [(obj.address + 16, size)]
        seen = 0
        while blocks and seen < nmsgs:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and seen < nmsgs:
                mtype, msize, flags = struct.unpack_from("<HHB", buf, p)
                body = p + 8
                seen += 1
                if flags & 0x02:
                    raise H5FormatError("shared object-header messages are not supported")
                if mtype == MSG_CONTINUATION:
                    blocks.append(struct.unpack_from("<QQ", buf, body))
                elif mtype == MSG_SYMBOL_TABLE:
                    obj._stab = struct.unpack_from("<QQ", buf, body)
                elif mtype == MSG_DATASPACE:
                    obj._shape = _decode_dataspace(buf, body)
                elif mtype == MSG_DATATYPE:
                    try:
                        obj._dtype = _decode_datatype(buf, body)
                    except H5FormatError as e:
                        obj._dtype_error = e
                elif mtype == MSG_LAYOUT:
                    obj._layout = self._decode_layout(body)
                elif mtype == MSG_ATTRIBUTE:
                    self._decode_attribute(obj, body)
                elif mtype == MSG_FILTER:
                    raise H5FormatError("filtered (compressed) datasets are not supported")
                elif mtype in (MSG_LINK, MSG_LINK_INFO, MSG_ATTR_INFO):
                    if mtype == MSG_LINK or self._dense(buf, body, mtype):
                        raise H5FormatError("new-style groups / dense attribute storage (libver='latest') are not supported")
                p = body + msize
        if obj._layout is not None and obj._dtype is None:
            raise getattr(obj, "_dtype_error", H5FormatError("dataset without a datatype"))

    @staticmethod
    def _dense(buf, body, mtype):
        # Link Info / Attribute Info messages with a defined fractal-heap address mean dense storage
        if mtype == MSG_LINK_INFO:
            flags = buf[body + 1]
            p = body + 2 + (8 if flags & 1 else 0)
        else:
            flags = buf[body + 1]
            p = body + 2 + (2 if flags & 1 else 0)
        return struct.unpack_from("<Q", buf, p)[0] != UNDEF

    def _decode_layout(self, body):
        version, cls = self.buf[body], self.buf[body + 1]
        if version != 3:
            raise H5FormatError(f"data layout message version {version} is not supported")
        if cls == 1:
            addr, size = struct.unpack_from("<QQ", self.buf, body + 2)
            return ("contiguous", addr, size)
        if cls == 0:
            size = struct.unpack_from("<H", self.buf, body + 2)[0]
            return ("compact", body + 4, size)
        raise H5FormatError("chunked datasets are not supported (Keras writes contiguous ones)")

    def _decode_attribute(self, obj, body):
        buf = self.buf
        version = buf[body]
        if version == 1:
            nlen, tlen, slen = struct.unpack_from("<HHH", buf, body + 2)
            p = body + 8
            name = buf[p:p + nlen].split(b"\0")[0].decode("utf-8")
            p += nlen + (-nlen % 8)
            tp = p
            p += tlen + (-tlen % 8)
            sp = p
            p += slen + (-slen % 8)
        elif version in (2, 3):
            nlen, tlen, slen = struct.unpack_from("<HHH", buf, body + 2)
            p = body + 8 + (1 if version == 3 else 0)
            name = buf[p:p + nlen].split(b"\0")[0].decode("utf-8")
            tp = p + nlen
            sp = tp + tlen
            p = sp + slen
        else:
            raise H5FormatError(f"attribute message version {version}")
        try:
            dtype = _decode_datatype(buf, tp)
        except H5FormatError:
            return                                  # e.g. a variable-length string attribute: not needed, skipped
        shape = _decode_dataspace(buf, sp)
        count = int(np.prod(shape)) if shape else 1
        obj.attrs[name] = np.frombuffer(buf[p:p + count * dtype.itemsize], dtype=dtype, count=count).reshape(shape).copy()

    # -- old-style groups ---------------------------------------------------------------------------
    def _group_members(self, btree_addr, heap_addr):
        buf = self.buf
        if buf[heap_addr:heap_addr + 4] != b"HEAP":
            raise H5FormatError("bad local-heap signature")
        heap_data = struct.unpack_from("<Q", buf, heap_addr + 24)[0]
        out = {}

        def name_at(off):
            start = heap_data + off
            return buf[start:buf.index(b"\0", start)].decode("utf-8")

        def walk(addr):
            if buf[addr:addr + 4] != b"TREE":
                raise H5FormatError("bad B-tree signature")
            ntype, level, used = struct.unpack_from("<BBH", buf, addr + 4)
            if ntype != 0:
                raise H5FormatError("not a group B-tree")
            p = addr + 24 + 8                       # skip key 0
            for _ in range(used):
                child = struct.unpack_from("<Q", buf, p)[0]
                p += 16                              # child address + next key
                if level > 0:
                    walk(child)
                    continue
                if buf[child:child + 4] != b"SNOD":
                    raise H5FormatError("bad symbol-table node signature")
                n = struct.unpack_from("<H", buf, child + 6)[0]
                for i in range(n):
                    name_off, header = struct.unpack_from("<QQ", buf, child + 8 + 40 * i)
                    out[name_at(name_off)] = header

        walk(btree_addr)
        return out


# ---------------------------------------------------------------------------------------------------------
# the Keras weight-file layout
# ---------------------------------------------------------------------------------------------------------
def _names_array(names):
    names = [n.encode("utf-8") if isinstance(n, str) else n for n in names]
    if not names:
        return np.zeros((0,), dtype=np.float64)     # what np.asarray([]) gives Keras for a layer without weights
    return np.array(names, dtype="S%d" % max(len(n) for n in names))


def _save_names(writer, node, key, names):
    """Keras' save_attributes_to_hdf5_group: one attribute, or `key0..keyN` chunks when it would exceed the header limit."""
    arr = _names_array(names)
    if arr.nbytes <= HEADER_LIMIT:
        writer.set_attr(node, key, arr)
        return
    nchunks = 1
    pieces = np.array_split(arr, nchunks)
    while any(p.nbytes > HEADER_LIMIT for p in pieces):
        nchunks += 1
        pieces = np.array_split(arr, nchunks)
    for i, p in enumerate(pieces):
        writer.set_attr(node, f"{key}{i}", p)


def _load_names(attrs, key):
    if key in attrs:
        a = attrs[key]
        return [] if a.dtype.kind != "S" else [n.decode("utf-8") for n in a.reshape(-1)]
    out, i = [], 0
    while f"{key}{i}" in attrs:
        out.extend(n.decode("utf-8") for n in attrs[f"{key}{i}"].reshape(-1))
        i += 1
    return out


TOP_LEVEL_GROUP = "top_level_model_weights"      # Keras >= 2.6: the model's own weights; a group that `layer_names` does not list


def save_keras_weights(path, layers, keras_version="2.5.0", backend="tensorflow"):
    """``layers``: [(layer_name, [(weight_name, array), ...]), ...] in ``model.layers`` order.  A group named
    ``top_level_model_weights`` is written like any other but left out of ``layer_names`` (Keras' convention for weights no layer
    owns: Keras 2.5's by-position loader never sees it, newer loaders and this package's read it)."""
    w = Writer(path)
    _save_names(w, w.root, "layer_names", [name for name, _ in layers if name != TOP_LEVEL_GROUP])
    w.set_attr(w.root, "backend", np.array(backend.encode("utf-8")))
    w.set_attr(w.root, "keras_version", np.array(keras_version.encode("utf-8")))
    for layer_name, weights in layers:
        g = w.group(layer_name)
        _save_names(w, g, "weight_names", [n for n, _ in weights])
        for weight_name, array in weights:
            w.create_dataset(layer_name + "/" + weight_name, np.asarray(array))
    w.close()


def load_keras_weights(path):
    """{weight_name: array} of a Keras HDF5 weight file (``model.save_weights``), or of the ``model_weights`` group of a
    full-model ``model.save`` file.  Follows ``layer_names`` / ``weight_names``; a file without them (not written by
    Keras) falls back to every dataset keyed by its path."""
    f = File(path)
    root = f.root
    if "layer_names" not in root.attrs and "layer_names0" not in root.attrs and "model_weights" in root.members():
        root = root["model_weights"]
    layer_names = _load_names(root.attrs, "layer_names")
    out = {}
    if not layer_names:
        for p, ds in root.visit_datasets().items():
            out[p] = ds.read()
        return out
    if TOP_LEVEL_GROUP in root.members() and TOP_LEVEL_GROUP not in layer_names:
        layer_names = layer_names + [TOP_LEVEL_GROUP]
    for layer in layer_names:
        g = root[layer]
        for wn in _load_names(g.attrs, "weight_names"):
            out[wn] = g[wn].read()
    return out
