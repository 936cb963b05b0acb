"""TensorFlow checkpoint container (wav2vec2/tfckpt.py): `<prefix>.index` string table + data shard, no TensorFlow.
What the reference's trainer loads and writes (src/main.py:132, src/training_utils.py:32-45).  No file written by TensorFlow
is available here (parity unpinned, see the module header): the byte layout is checked against a hand-assembled known answer
and against google.protobuf's own serializer for the three messages, the rest by round trips and corruption tests."""
import os
import struct

import numpy as np
import pytest

from wav2vec2 import tfckpt as K
from wav2vec2 import tfrecord as T


def _bundle_protos():
    """BundleHeaderProto / BundleEntryProto / TrackableObjectGraph message classes from descriptors transcribed from the
    published tensor_bundle.proto, versions.proto, tensor_shape.proto and trackable_object_graph.proto field numbers."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="w2v2_bundle_subset.proto", package="w2v2b", syntax="proto3")
    F = descriptor_pb2.FieldDescriptorProto
    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED

    def msg(name, fields, nested=()):
        m = descriptor_pb2.DescriptorProto(name=name)
        for fname, num, typ, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=typ, label=label)
            if tname:
                f.type_name = tname
        for n in nested:
            m.nested_type.add().CopyFrom(n)
        return m

    fd.message_type.add().CopyFrom(msg("VersionDef", [("producer", 1, F.TYPE_INT32, OPT, None), ("min_consumer", 2, F.TYPE_INT32, OPT, None)]))
    fd.message_type.add().CopyFrom(msg("BundleHeaderProto", [("num_shards", 1, F.TYPE_INT32, OPT, None), ("endianness", 2, F.TYPE_INT32, OPT, None),
                                                             ("version", 3, F.TYPE_MESSAGE, OPT, ".w2v2b.VersionDef")]))
    dim = msg("Dim", [("size", 1, F.TYPE_INT64, OPT, None), ("name", 2, F.TYPE_STRING, OPT, None)])
    fd.message_type.add().CopyFrom(msg("TensorShapeProto", [("dim", 2, F.TYPE_MESSAGE, REP, ".w2v2b.TensorShapeProto.Dim")], nested=[dim]))
    fd.message_type.add().CopyFrom(msg("BundleEntryProto", [("dtype", 1, F.TYPE_INT32, OPT, None),
                                                            ("shape", 2, F.TYPE_MESSAGE, OPT, ".w2v2b.TensorShapeProto"),
                                                            ("shard_id", 3, F.TYPE_INT32, OPT, None), ("offset", 4, F.TYPE_INT64, OPT, None),
                                                            ("size", 5, F.TYPE_INT64, OPT, None), ("crc32c", 6, F.TYPE_FIXED32, OPT, None)]))
    ref = msg("ObjectReference", [("node_id", 1, F.TYPE_INT32, OPT, None), ("local_name", 2, F.TYPE_STRING, OPT, None)])
    ser = msg("SerializedTensor", [("name", 1, F.TYPE_STRING, OPT, None), ("full_name", 2, F.TYPE_STRING, OPT, None),
                                   ("checkpoint_key", 3, F.TYPE_STRING, OPT, None)])
    obj = msg("TrackableObject", [("children", 1, F.TYPE_MESSAGE, REP, ".w2v2b.TrackableObjectGraph.TrackableObject.ObjectReference"),
                                  ("attributes", 2, F.TYPE_MESSAGE, REP, ".w2v2b.TrackableObjectGraph.TrackableObject.SerializedTensor")],
              nested=[ref, ser])
    fd.message_type.add().CopyFrom(msg("TrackableObjectGraph", [("nodes", 1, F.TYPE_MESSAGE, REP, ".w2v2b.TrackableObjectGraph.TrackableObject")],
                                       nested=[obj]))
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    cls = (lambda n: get(pool.FindMessageTypeByName("w2v2b." + n))) if get else \
          (lambda n: message_factory.MessageFactory(pool).GetPrototype(pool.FindMessageTypeByName("w2v2b." + n)))
    return cls("BundleHeaderProto"), cls("BundleEntryProto"), cls("TrackableObjectGraph")


def test_crc32c_extend_matches_one_shot_and_native_matches_python():
    rs = np.random.RandomState(0)
    blob = rs.randint(0, 256, size=100003).astype(np.uint8).tobytes()
    whole = T.crc32c(blob)
    assert K._crc32c(blob) == whole                                   # (native slicing-by-8 when the library is built)
    for cut in (0, 1, 7, 8, 9, 4096, len(blob)):
        assert K._crc32c(blob[cut:], K._crc32c(blob[:cut])) == whole
    assert K._crc32c(b"123456789") == 0xE3069283
    saved, K._NATIVE_CRC = K._NATIVE_CRC, False                       # the byte-at-a-time fallback is the same function
    try:
        assert K._crc32c(blob[:5000]) == T.crc32c(blob[:5000])
        assert K._crc32c(blob[100:5000], K._crc32c(blob[:100])) == T.crc32c(blob[:5000])
    finally:
        K._NATIVE_CRC = saved
    assert K._mask(0) == 0xA282EAD8


def test_bundle_messages_match_protobuf():
    Header, Entry, Graph = _bundle_protos()
    h = Header(num_shards=1)
    h.version.producer = 1
    assert K._header_proto() == h.SerializeToString(deterministic=True) == bytes.fromhex("08011a020801")
    for dtype, shape, off, size, crc in ((1, (768, 3072), 4096, 768 * 3072 * 4, 0xDEADBEEF), (1, (), 0, 4, 1), (3, (5,), 12, 20, 0),
                                         (1, (0, 3), 7, 0, 5)):
        e = Entry(dtype=dtype, offset=off, size=size, crc32c=crc)
        e.shape.SetInParent()
        for d in shape:
            e.shape.dim.add(size=d)
        assert K._entry_proto(dtype, shape, off, size, crc) == e.SerializeToString(deterministic=True), (shape, off)
        back = K._parse_entry(e.SerializeToString())
        assert (back["dtype"], tuple(back["shape"]), back["offset"], back["size"], back["crc32c"]) == (dtype, shape, off, size, crc)
    names = ["a/kernel", "a/bias", "b/c/gamma"]
    blob, keys = K._object_graph_for(sorted(names))
    g = Graph.FromString(blob)
    assert g.SerializeToString(deterministic=True) == blob
    assert [c.local_name for c in g.nodes[0].children] == ["a", "b"]
    got = {a.full_name: a.checkpoint_key for n in g.nodes for a in n.attributes}
    assert got == keys == {n: n + "/.ATTRIBUTES/VARIABLE_VALUE" for n in names}
    # a graph as TensorFlow writes it (attribute paths unrelated to the variable names) maps through full_name
    tf_like = Graph()
    root = tf_like.nodes.add()
    root.children.add(node_id=1, local_name="layer_with_weights-0")
    leaf = tf_like.nodes.add()
    leaf.attributes.add(name="VARIABLE_VALUE", full_name="wav2vec2-ctc/lm_head/kernel", checkpoint_key="layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE")
    leaf.attributes.add(name="OTHER", full_name="x", checkpoint_key="y")
    assert K._object_graph_names(tf_like.SerializeToString()) == {"wav2vec2-ctc/lm_head/kernel": "layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE"}


def test_one_entry_table_known_answer(tmp_path):
    """The index of a bundle with no tensors, assembled by hand from the table format: one data block holding the entry
    ("" -> header), restart array [0], count 1; the empty metaindex block; an index block whose single entry is the short
    successor of the last key ("" stays "") -> handle(0, 17); footer.  Every block is followed by type 0 + masked CRC-32C."""
    header = bytes.fromhex("08011a020801")
    data = bytes([0, 0, len(header)]) + header + struct.pack("<II", 0, 1)
    assert len(data) == 17

    def trailer(block):
        return b"\x00" + struct.pack("<I", T.masked_crc32c(block + b"\x00"))

    meta = struct.pack("<II", 0, 1)
    meta_off = len(data) + 5
    idx_entry_value = bytes([0, 17])                                   # varint offset 0, varint size 17
    idx = bytes([0, 0, len(idx_entry_value)]) + idx_entry_value + struct.pack("<II", 0, 1)
    idx_off = meta_off + len(meta) + 5
    footer = bytes([meta_off, len(meta), idx_off, len(idx)])
    footer += bytes(40 - len(footer)) + bytes.fromhex("57fb808b247547db")
    want = data + trailer(data) + meta + trailer(meta) + idx + trailer(idx) + footer
    path = str(tmp_path / "empty.index")
    K.write_table(path, [(b"", header)])
    assert open(path, "rb").read() == want
    assert K.read_table(path) == {b"": header}


def test_table_roundtrip_restarts_blocks_and_corruption(tmp_path):
    rs = np.random.RandomState(1)
    keys = sorted({("encoder/layers/%d/%s/%s" % (rs.randint(24), rs.choice(["attention/q_proj", "feed_forward/output_dense", "ln"]),
                                                 rs.choice(["kernel", "bias", "gamma", "beta"]))).encode() + bytes([rs.randint(48, 58)])
                   for _ in range(400)} | {b"", b"\xff\xff", b"\xff\xff\x00"})
    items = [(k, rs.randint(0, 256, size=rs.randint(0, 90)).astype(np.uint8).tobytes()) for k in keys]
    for block_size in (K.BLOCK_SIZE, 512, 64):                          # one block / a few / nearly one entry per block
        path = str(tmp_path / f"t{block_size}.index")
        K.write_table(path, items, block_size=block_size)
        got = K.read_table(path)
        assert list(got.items()) == items
    raw = bytearray(open(path, "rb").read())
    raw[10] ^= 0x40
    open(path, "wb").write(raw)
    with pytest.raises(ValueError, match="checksum"):
        K.read_table(path)
    raw[10] ^= 0x40
    raw[-1] ^= 1
    open(path, "wb").write(raw)
    with pytest.raises(ValueError, match="magic"):
        K.read_table(path)
    with pytest.raises(ValueError, match="increasing"):
        K.write_table(path, [(b"b", b""), (b"a", b"")])
    # separators as the table builder shortens them
    assert K._shortest_separator(b"abcdef", b"abzzzz") == b"abd"
    assert K._shortest_separator(b"abc", b"abcd") == b"abc"             # a prefix of the limit: unchanged
    assert K._shortest_separator(b"ab\xff", b"ac") == b"ab\xff"
    assert K._short_successor(b"\xff\xffa") == b"\xff\xffb" and K._short_successor(b"\xff") == b"\xff" and K._short_successor(b"") == b""


@pytest.mark.parametrize("object_graph", [False, True])
def test_checkpoint_roundtrip(tmp_path, object_graph):
    rs = np.random.RandomState(2)
    tensors = {"wav2vec2-ctc/wav2vec2/encoder/layers/%d/attention/q_proj/kernel" % i: rs.randn(8, 8).astype(np.float32) for i in range(20)}
    tensors["wav2vec2-ctc/lm_head/bias"] = rs.randn(32).astype(np.float32)
    tensors["wav2vec2-ctc/wav2vec2/masked_spec_embed"] = rs.randn(8).astype(np.float32)
    tensors["scalar"] = np.float32(2.5)
    tensors["steps"] = np.array([3, 4], np.int64)
    tensors["empty"] = np.zeros((0, 4), np.float32)
    prefix = str(tmp_path / "run_stage1" / "tf_model")
    K.write_checkpoint(prefix, tensors, object_graph=object_graph)
    assert sorted(os.listdir(tmp_path / "run_stage1")) == ["tf_model.data-00000-of-00001", "tf_model.index"]
    assert K.is_checkpoint(prefix) and not K.is_checkpoint(prefix + "x")
    back = K.read_checkpoint(prefix)
    assert set(back) == set(tensors)
    for n, a in tensors.items():
        assert back[n].dtype == np.asarray(a).dtype and back[n].shape == np.asarray(a).shape and np.array_equal(back[n], a), n
    r = K.BundleReader(prefix)
    assert r.header == dict(num_shards=1, endianness=0, producer=1)
    if object_graph:
        assert "_CHECKPOINTABLE_OBJECT_GRAPH" in r.keys()
        assert all(k.endswith("/.ATTRIBUTES/VARIABLE_VALUE") for k in r.keys() if not k.startswith("_"))
        assert r.variables()["scalar"] == "scalar/.ATTRIBUTES/VARIABLE_VALUE"
    else:
        assert sorted(r.keys(), key=str.encode) == sorted(tensors, key=str.encode)
    # tensors lie back to back in key order, and the sizes add up to the shard
    ents = sorted(r.entries.items(), key=lambda kv: kv[0].encode())
    off = 0
    for _, e in ents:
        assert e["offset"] == off
        off += e["size"]
    assert off == os.path.getsize(prefix + ".data-00000-of-00001")
    # a flipped byte in the shard is caught by the tensor's checksum
    key = [k for k, e in ents if e["shape"] == [8, 8]][0]
    with open(prefix + ".data-00000-of-00001", "r+b") as f:
        f.seek(r.entries[key]["offset"] + 5)
        b = f.read(1)
        f.seek(-1, 1)
        f.write(bytes([b[0] ^ 0x10]))
    with pytest.raises(ValueError, match="checksum"):
        K.BundleReader(prefix).tensor(key)
    assert K.BundleReader(prefix, check_crc=False).tensor(key).shape == (8, 8)
    with pytest.raises(FileNotFoundError):
        K.read_checkpoint(str(tmp_path / "nothing" / "tf_model"))


def test_bfloat16_and_half_variables_read_as_values(tmp_path):
    """A mixed-precision run may hold bfloat16 variables: they come back as float32 values."""
    prefix = str(tmp_path / "tf_model")
    vals = np.array([1.0, -2.5, 3.140625, 0.0], np.float32)
    K.write_checkpoint(prefix, {"v": vals.astype(np.float16)})
    assert np.array_equal(K.read_checkpoint(prefix)["v"], vals.astype(np.float16))
    # re-tag the entry as DT_BFLOAT16 with bf16 bit patterns on disk
    bits = (vals.view(np.uint32) >> 16).astype("<u2")
    open(prefix + ".data-00000-of-00001", "wb").write(bits.tobytes())
    K.write_table(prefix + ".index", [(b"", K._header_proto()),
                                      (b"v", K._entry_proto(K.DT_BFLOAT16, (4,), 0, 8, K._mask(K._crc32c(bits.tobytes()))))])
    got = K.read_checkpoint(prefix)["v"]
    assert got.dtype == np.float32 and np.array_equal(got, vals)
