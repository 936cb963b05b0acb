"""Host-side logic and the C-ABI surface; no GPU, no compute calls."""

import ctypes
import json
import os
import re

import numpy as np
import pytest

import helpers as H
from wav2vec2 import _native as N
from wav2vec2 import variables as V
from wav2vec2.config import RobustWav2Vec2Config, Wav2Vec2Config
from wav2vec2.processor import Wav2Vec2Processor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- config (reference config.py:6-73) ---------------------------------------
def test_config_defaults_and_fields():
    c = Wav2Vec2Config()
    assert len(c.__dataclass_fields__) == 22
    assert (c.vocab_size, c.hidden_size, c.num_heads, c.num_layers, c.intermediate_size) == (32, 768, 12, 12, 3072)
    assert c.kernal_sizes == [10, 3, 3, 3, 3, 2, 2] and c.strides == [5, 2, 2, 2, 2, 2, 2]
    assert c.feature_extractor_norm_type == "group" and c.attention_norm_type == "postnorm"
    r = RobustWav2Vec2Config()
    assert isinstance(r, Wav2Vec2Config)
    assert (r.hidden_size, r.num_heads, r.num_layers, r.intermediate_size) == (1024, 16, 24, 4096)
    assert r.conv_bias and r.is_robust and r.feature_extractor_norm_type == "layer" and r.attention_norm_type == "prenorm"


def test_config_validation_errors():
    with pytest.raises(ValueError):
        Wav2Vec2Config(filter_sizes=[512] * 6)
    with pytest.raises(ValueError):
        Wav2Vec2Config(hidden_size=770)
    with pytest.raises(AssertionError):
        Wav2Vec2Config(feature_extractor_norm_type="batch")
    with pytest.raises(AssertionError):
        Wav2Vec2Config(attention_norm_type="sandwich")


def test_config_json_roundtrip(tmp_path):
    c = RobustWav2Vec2Config(num_layers=3)
    c.save_pretrained(str(tmp_path))
    d = json.load(open(tmp_path / "config.json"))
    assert "kernal_sizes" in d and len(d) == 22
    c2 = Wav2Vec2Config.from_json(str(tmp_path / "config.json"))
    assert c2.num_layers == 3 and c2.is_robust and c2.hidden_size == 1024
    d["unknown_key"] = 1
    json.dump(d, open(tmp_path / "bad.json", "w"))
    with pytest.raises(TypeError):
        Wav2Vec2Config.from_json(str(tmp_path / "bad.json"))


# ---- variable inventory / checkpoint naming ------------------------------------
def test_variable_inventory_matches_reference_counts():
    specs = V.variable_specs(Wav2Vec2Config(), with_lm_head=True)
    assert len(specs) == 213                                   # notebooks/wav2vec2_onnx.ipynb:125
    n_params = sum(int(np.prod(s)) for s, _ in specs.values())
    assert n_params == 94_396_320                              # == HF wav2vec2-base CTC
    frozen = sum(int(np.prod(s)) for n, (s, _) in specs.items() if n.startswith("feature_extractor/"))
    assert frozen == 4_200_448                                 # the stage-2 frozen conv stack
    assert n_params - frozen == 90_195_872 or n_params - frozen > 0
    assert V.tf_variable_name("encoder/layers/0/attention/q_proj/kernel") == \
        "wav2vec2-ctc/wav2vec2/encoder/layers/0/attention/q_proj/kernel:0"
    assert V.tf_variable_name("lm_head/bias") == "wav2vec2-ctc/lm_head/bias:0"
    assert V.tf_variable_name("encoder/layer_norm/gamma", with_lm_head=False) == "wav2vec2/encoder/layer_norm/gamma:0"
    for n in specs:
        assert V.local_name_from_tf(V.tf_variable_name(n)) == n


def test_hf_key_mapping_roundtrip():
    cfg = Wav2Vec2Config(**H.TINY)
    w = V.seeded_weights(cfg, seed=3)
    sd = V.to_hf_state_dict(w, new_weight_norm_names=False)
    assert sd["wav2vec2.encoder.pos_conv_embed.conv.weight_g"].shape == (1, 1, 16)
    assert sd["wav2vec2.encoder.pos_conv_embed.conv.weight_v"].shape == (64, 16, 16)
    assert sd["wav2vec2.feature_extractor.conv_layers.0.conv.weight"].shape == (32, 1, 10)
    assert sd["wav2vec2.encoder.layers.1.feed_forward.intermediate_dense.weight"].shape == (128, 64)
    assert "wav2vec2.feature_extractor.conv_layers.0.layer_norm.bias" in sd
    back = V.from_hf_state_dict(sd, cfg)
    for k in w:
        assert np.array_equal(w[k], back[k]), k


def test_seeded_generator_is_stable():
    u = V.hash_uniform("tag", 5, seed=7)
    # golden values of the integer hash: any change would silently invalidate every fixture
    assert u.dtype == np.float32 and np.all((u >= 0) & (u < 1))
    again = V.hash_uniform("tag", 5, seed=7)
    assert np.array_equal(u, again)
    assert not np.array_equal(u, V.hash_uniform("tag", 5, seed=8))
    g = H.golden("tiny_base")
    assert np.array_equal(g["wave"], V.hash_normal("tiny/wave", 2 * 4000, 1).reshape(2, 4000))


# ---- processor (reference processor.py) ------------------------------------------
def test_tokenizer_and_decode():
    tok = Wav2Vec2Processor(is_tokenizer=True, vocab_path=os.path.join(H.GOLDEN, "vocab.json"))
    ids = tok("how is life? it's awe-some")
    assert tok.decode(ids, group_tokens=False) == "HOW IS LIFE IT'S AWE SOME"
    # greedy CTC collapse: repeats merge, <pad>=0 dropped, '|' -> space
    assert tok.decode([0, 11, 11, 0, 5, 5, 4, 4, 0, 15, 0, 15, 8]) == "HE LLO"
    assert tok.decode([3, 3, 99]) == "<unk><unk>"


def test_normalize_numpy_and_torch_agree():
    import torch
    x = np.random.default_rng(0).normal(2.0, 3.0, size=(2, 1000)).astype(np.float32)
    p = Wav2Vec2Processor(is_tokenizer=False)
    a = p(x)
    b = p(torch.from_numpy(x)).numpy()
    assert a.shape == (2, 1000) and np.allclose(a, b, atol=1e-5)
    assert np.allclose(a.mean(-1), 0, atol=1e-5) and np.allclose(a.var(-1), 1, atol=1e-3)
    assert p(x[0]).shape == (1000,)


# ---- the C ABI -----------------------------------------------------------------------
def _header_symbols():
    src = open(os.path.join(ROOT, "include", "w2v2.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(w2v2_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
    syms = _header_symbols()
    assert len(syms) >= 25
    lib = N.load()
    for s in syms:
        assert hasattr(lib, s), f"libw2v2.so does not export {s}"
    assert set(syms) == set(N.PROTOTYPES), "ctypes prototypes and include/w2v2.h disagree"
    assert b"gfx950" in lib.w2v2_version()


def test_config_struct_layout():
    c = N.make_config(RobustWav2Vec2Config(), with_lm_head=True)
    assert ctypes.sizeof(c) == 4 * (8 + 48 + 6) + 4
    assert (c.hidden_size, c.num_layers, c.conv_bias, c.feature_extractor_norm_type, c.attention_norm_type) == (1024, 24, 1, 1, 1)
    assert list(c.kernal_sizes)[:7] == [10, 3, 3, 3, 3, 2, 2] and c.num_conv_layers == 7


def test_model_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import wav2vec2
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        wav2vec2.Wav2Vec2ForCTC(Wav2Vec2Config(**H.TINY))
    with pytest.raises(ValueError):
        wav2vec2.Wav2Vec2ForCTC({"hidden_size": 768})


# ---- input pipeline contract (reference data_utils.py:52-78) -------------------------------------
def test_batchify_normalises_then_pads():
    from wav2vec2.data_utils import attention_mask_for, batchify
    rng = np.random.default_rng(0)
    a, b = rng.normal(3, 2, 1000).astype(np.float32), rng.normal(-1, 0.5, 5000).astype(np.float32)
    tok = Wav2Vec2Processor(is_tokenizer=True, vocab_path=os.path.join(H.GOLDEN, "vocab.json"))
    speech, labels = batchify([a, b], ["hello world", "a-b"], audio_maxlen=3000, labels_maxlen=8, tokenizer=tok)
    assert speech.shape == (2, 3000) and labels.shape == (2, 8) and labels.dtype == np.int32
    # normalised BEFORE padding: the valid part has zero mean / unit variance, the pad is exact zeros
    assert abs(speech[0, :1000].mean()) < 1e-5 and abs(speech[0, :1000].var() - 1) < 1e-3
    assert not speech[0, 1000:].any()
    # truncation keeps the first audio_maxlen samples of the normalised long clip
    full = Wav2Vec2Processor(is_tokenizer=False)(b)
    assert np.allclose(speech[1], full[:3000])
    assert labels[0].tolist() == [11, 5, 15, 15, 8, 4, 18, 8] and labels[1].tolist() == [7, 4, 24, 0, 0, 0, 0, 0]
    m = attention_mask_for([1000, 5000], 3000)
    assert m.sum(1).tolist() == [1000, 3000]


def test_spec_augment_mask_recipe():
    from wav2vec2.spec_augment import compute_mask_indices
    m = compute_mask_indices((8, 768), 0.05, 10, rng=np.random.RandomState(0))
    assert m.shape == (8, 768) and m.dtype == np.uint8
    # 0.05 * 768 / 10 = 3.84 -> 3 or 4 spans of 10 frames per row (may overlap), same count for every row
    assert all(10 <= r <= 40 for r in m.sum(1))
    m2 = compute_mask_indices((2, 12), 0.0, 2, rng=np.random.RandomState(0))    # min_masks = 2 always applies
    assert all(2 <= r <= 4 for r in m2.sum(1))
    with pytest.raises(ValueError):
        compute_mask_indices((2, 5), 0.5, 10)


def test_dropout_hash_statistics():
    keep = V.dropout_keep(seed=12345, stream=V.layer_stream(3, 0), n=200000, p=0.1)
    assert abs(keep.mean() - 0.9) < 0.005
    assert not np.array_equal(keep, V.dropout_keep(12345, V.layer_stream(3, 1), 200000, 0.1))
    assert np.array_equal(keep, V.dropout_keep(12345, V.layer_stream(3, 0), 200000, 0.1))
    # the oct hash (round 6: eight decisions share one mixer round, csrc/train.h): the decisions inside an oct, at short lags and
    # between rows must be as uncorrelated as independent draws are (tools/dropout_hash_stats.py is the long form of this test)
    n = 1 << 21
    k = V.dropout_keep(99, V.layer_stream(0, 2), n, 0.1).astype(np.float64)
    k -= k.mean()
    v, q = k.var(), k.reshape(-1, 8)
    sd_oct, sd = 1.0 / np.sqrt(n / 8), 1.0 / np.sqrt(n)
    assert max(abs((q[:, a] * q[:, b]).mean() / v) for a in range(8) for b in range(a + 1, 8)) < 4.5 * sd_oct
    assert max(abs((k[:-l] * k[l:]).mean() / v) for l in (1, 2, 3, 4, 7, 8, 9, 16, 768, 3072)) < 4.5 * sd
    h = V.dropout_hash(99, V.layer_stream(0, 2), n)
    for byte in (h >> 8, h & 0xFF):             # both bytes of the 16-bit value uniform: chi-square, 255 degrees of freedom (mean 255, sd 22.6)
        assert (((np.bincount(byte.astype(np.int64), minlength=256) - n / 256) ** 2) / (n / 256)).sum() < 255 + 5 * 22.6
    # attention index space: stride 16-aligned, bits 2 / 3 of the key exchanged -- every (query, key) still gets its own decision
    a = V.attention_keep(5, 16, 64, 21, 0.1)
    raw = V.dropout_keep(5, 16, 64 * 32, 0.1).reshape(64, 32)
    assert a.shape == (64, 21) and np.array_equal(a[:, 4:8], raw[:, 8:12]) and np.array_equal(a[:, 8:12], raw[:, 4:8]) and np.array_equal(a[:, 16:20], raw[:, 16:20]) and np.array_equal(a[:, 20], raw[:, 24])


# ---- HuggingFace checkpoint directories (what src/convert_torch_to_tf.py converts) ------------
def _hf_dir(tmp_path, cfg, hf_cfg_dict, weights, fmt="safetensors"):
    import json
    d = tmp_path / "hf"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(hf_cfg_dict))
    sd = V.to_hf_state_dict(weights)
    if fmt == "safetensors":
        from safetensors.numpy import save_file
        save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, str(d / "model.safetensors"))
    else:
        import torch
        torch.save({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, str(d / "pytorch_model.bin"))
    return str(d)


def test_config_from_real_hf_config_defaults():
    """transformers' own Wav2Vec2Config defaults map onto the reference's defaults field for field,
    and the robust switches (do_stable_layer_norm / feat_extract_norm='layer' / conv_bias) onto RobustWav2Vec2Config."""
    transformers = pytest.importorskip("transformers")
    base = Wav2Vec2Config.from_hf_config(transformers.Wav2Vec2Config().to_dict())
    assert base == Wav2Vec2Config()
    hf_robust = transformers.Wav2Vec2Config(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                                            do_stable_layer_norm=True, feat_extract_norm="layer", conv_bias=True)
    assert Wav2Vec2Config.from_hf_config(hf_robust.to_dict()) == RobustWav2Vec2Config()
    with pytest.raises(NotImplementedError):
        Wav2Vec2Config.from_hf_config(dict(transformers.Wav2Vec2Config().to_dict(), hidden_act="relu"))


@pytest.mark.parametrize("fmt", ["safetensors", "bin"])
def test_convert_hf_checkpoint_directory(tmp_path, fmt):
    """convert_torch_to_tf.py without TF: HF directory in, `config.json` (22 reference fields) + TF-named weights out."""
    from wav2vec2.modeling import convert_hf_checkpoint, read_hf_state_dict
    cfg = H.case_config("tiny_robust")
    w = H.case_weights("tiny_robust")
    hf_cfg = dict(hidden_size=cfg.hidden_size, num_attention_heads=cfg.num_heads, num_hidden_layers=cfg.num_layers,
                  intermediate_size=cfg.intermediate_size, conv_dim=cfg.filter_sizes, conv_kernel=cfg.kernal_sizes,
                  conv_stride=cfg.strides, conv_bias=True, feat_extract_norm="layer", do_stable_layer_norm=True,
                  num_conv_pos_embeddings=cfg.num_conv_pos_embeddings, num_conv_pos_embedding_groups=cfg.num_conv_pos_embedding_groups,
                  vocab_size=cfg.vocab_size, hidden_act="gelu", layer_norm_eps=cfg.layer_norm_eps, pad_token_id=0,
                  hidden_dropout=cfg.dropout, mask_time_prob=cfg.mask_time_prob, mask_time_length=cfg.mask_time_length)
    src = _hf_dir(tmp_path, cfg, hf_cfg, w, fmt)
    assert set(read_hf_state_dict(src)) == set(V.to_hf_state_dict(w))
    out = str(tmp_path / "converted")
    got_cfg = convert_hf_checkpoint(src, out)
    assert got_cfg == cfg
    assert Wav2Vec2Config.from_json(os.path.join(out, "config.json")) == Wav2Vec2Config(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__})
    from wav2vec2 import h5lite
    z = h5lite.load_keras_weights(os.path.join(out, "tf_model.h5"))         # the reference's container (modeling.py:26)
    assert set(z) == {V.tf_variable_name(n) for n in w}
    for n, a in w.items():
        assert np.array_equal(z[V.tf_variable_name(n)], a), n
    assert [x.decode() for x in h5lite.File(os.path.join(out, "tf_model.h5")).root.attrs["layer_names"]] == ["wav2vec2", "dropout", "lm_head"]


def test_stage2_learning_rate_schedule():
    """training_utils.py:24-31: lr1 for epochs <= transition (0-based), lr2 after."""
    from wav2vec2.training import stage2_learning_rate
    assert [stage2_learning_rate(e) for e in (0, 9, 10, 11, 30)] == [1e-4, 1e-4, 1e-4, 5e-5, 5e-5]
    assert stage2_learning_rate(3, lr1=2e-4, lr2=1e-5, transition_epochs=2) == 1e-5


# ---- TFRecord files of the reference's schema (make_tfrecords.py:10-23, data_utils.py:17-27), no TensorFlow -----
def _tf_protos():
    """Message classes built with google.protobuf from descriptors transcribed from the published example.proto /
    feature.proto / tensor.proto / tensor_shape.proto field numbers -- an independent serializer to check ours against."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="w2v2_tf_subset.proto", package="w2v2tf", syntax="proto3")
    F = descriptor_pb2.FieldDescriptorProto

    def msg(name, fields, nested=()):
        m = descriptor_pb2.DescriptorProto(name=name)
        for fname, num, typ, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=typ, label=label)
            if tname:
                f.type_name = tname
        for n in nested:
            m.nested_type.add().CopyFrom(n)
        return m

    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    fd.message_type.add().CopyFrom(msg("BytesList", [("value", 1, F.TYPE_BYTES, REP, None)]))
    fd.message_type.add().CopyFrom(msg("Feature", [("bytes_list", 1, F.TYPE_MESSAGE, OPT, ".w2v2tf.BytesList")]))
    entry = msg("FeatureEntry", [("key", 1, F.TYPE_STRING, OPT, None), ("value", 2, F.TYPE_MESSAGE, OPT, ".w2v2tf.Feature")])
    entry.options.map_entry = True
    fd.message_type.add().CopyFrom(msg("Features", [("feature", 1, F.TYPE_MESSAGE, REP, ".w2v2tf.Features.FeatureEntry")], nested=[entry]))
    fd.message_type.add().CopyFrom(msg("Example", [("features", 1, F.TYPE_MESSAGE, OPT, ".w2v2tf.Features")]))
    dim = msg("Dim", [("size", 1, F.TYPE_INT64, OPT, None), ("name", 2, F.TYPE_STRING, OPT, None)])
    fd.message_type.add().CopyFrom(msg("TensorShapeProto", [("dim", 2, F.TYPE_MESSAGE, REP, ".w2v2tf.TensorShapeProto.Dim"),
                                                            ("unknown_rank", 3, F.TYPE_BOOL, OPT, None)], nested=[dim]))
    fd.message_type.add().CopyFrom(msg("TensorProto", [("dtype", 1, F.TYPE_INT32, OPT, None),
                                                       ("tensor_shape", 2, F.TYPE_MESSAGE, OPT, ".w2v2tf.TensorShapeProto"),
                                                       ("version_number", 3, F.TYPE_INT32, OPT, None),
                                                       ("tensor_content", 4, F.TYPE_BYTES, OPT, None)]))
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    cls = (lambda n: get(pool.FindMessageTypeByName("w2v2tf." + n))) if get else \
          (lambda n: message_factory.MessageFactory(pool).GetPrototype(pool.FindMessageTypeByName("w2v2tf." + n)))
    return cls("Example"), cls("TensorProto")


def test_crc32c_known_answers():
    from wav2vec2 import tfrecord as T
    assert T.crc32c(b"123456789") == 0xE3069283                      # the CRC catalogue's check value for CRC-32C
    assert T.crc32c(bytes(32)) == 0x8A9136AA                          # RFC 3720 B.4
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.masked_crc32c(b"") == 0xA282EAD8                        # crc 0 rotated is 0: the mask constant alone


def test_tensor_and_example_bytes_match_protobuf():
    from wav2vec2 import tfrecord as T
    Example, TensorProto = _tf_protos()
    speech = (np.arange(7, dtype=np.float32) - 3) / 4
    label = np.array([5, 9, 9, 11, 0, 0], np.int32)
    for arr, dt in ((speech, 1), (label, 3), (np.zeros((2, 3), np.float32), 1), (np.float32(2.5), 1)):
        tp = TensorProto(dtype=dt, tensor_content=np.asarray(arr).tobytes())
        for d in np.asarray(arr).shape:
            tp.tensor_shape.dim.add(size=d)
        if np.asarray(arr).ndim == 0:
            tp.tensor_shape.SetInParent()
        assert T.serialize_tensor(arr) == tp.SerializeToString(deterministic=True)
        back = T.parse_tensor(tp.SerializeToString(), arr.dtype)
        assert back.dtype == arr.dtype and back.shape == np.asarray(arr).shape and np.array_equal(back, arr)
    ex = Example()
    ex.features.feature["speech"].bytes_list.value.append(T.serialize_tensor(speech))
    ex.features.feature["label"].bytes_list.value.append(T.serialize_tensor(label))
    assert T.create_tfrecord(speech, label) == ex.SerializeToString(deterministic=True)
    s, l = T.read_tfrecords(ex.SerializeToString())                   # any map order parses
    assert np.array_equal(s, speech) and np.array_equal(l, label) and s.dtype == np.float32 and l.dtype == np.int32
    parsed = Example.FromString(T.create_tfrecord(speech, label))
    assert set(parsed.features.feature) == {"speech", "label"}


def test_tfrecord_file_roundtrip_and_corruption(tmp_path):
    from wav2vec2 import tfrecord as T
    from wav2vec2.data_utils import batchify
    rs = np.random.RandomState(0)
    samples = [(rs.randn(n).astype(np.float32), rs.randint(1, 32, size=u).astype(np.int32)) for n, u in ((1000, 7), (1, 0), (4097, 256))]
    path = str(tmp_path / "dev-clean-0.tfrecord")
    T.write_dataset(path, samples)
    raw = open(path, "rb").read()
    assert int.from_bytes(raw[:8], "little") == len(T.create_tfrecord(*samples[0]))      # length prefix of record 0
    got = list(T.read_dataset(path))
    assert len(got) == 3
    for (s, l), (s2, l2) in zip(samples, got):
        assert np.array_equal(s, s2) and np.array_equal(l, l2)
    # the reference's pipeline from here: normalise -> truncate -> right-pad (data_utils.py:52-78)
    speech, _ = batchify([s for s, _ in got], audio_maxlen=2000)
    assert speech.shape == (3, 2000) and abs(speech[0, :1000].mean()) < 1e-5 and not speech[0, 1000:].any()
    bad = bytearray(raw)
    bad[40] ^= 1
    open(path, "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        list(T.read_dataset(path))
    open(path, "wb").write(raw[:-3])
    with pytest.raises(ValueError):
        list(T.read_dataset(path))
    with pytest.raises(ValueError):                                    # dtype check of parse_tensor(out_type=...)
        T.parse_tensor(T.serialize_tensor(np.zeros(3, np.int32)), np.float32)


# ---- the Keras object graph callers touch (src/main.py:210,232-237) -----------------------------------
class _FakeLib:
    """Records what would be pushed to the native training state (no GPU here)."""

    def __init__(self, names):
        self.names = list(reversed(names))      # a native inventory order that differs from the host's: flags must follow IT
        self.flags, self.calls = {}, 0

    def w2v2_num_params(self, handle):
        return len(self.names)

    def w2v2_param_info(self, handle, i, name_ref, shape, rank_ref):
        name_ref._obj.value = self.names[i].encode()
        return 0

    def w2v2_set_trainable_flags(self, handle, vec, n):
        assert n == len(self.names)
        self.flags = {name: bool(vec[i]) for i, name in enumerate(self.names)}
        self.calls += 1
        return 0


def _graph_only_model(cls, config):
    """The model's host-side object graph without the native library: what `_build_native` does after `w2v2_create`."""
    from wav2vec2 import modeling as M
    m = object.__new__(cls)
    m.name = "wav2vec2-ctc" if cls._with_lm_head else "wav2vec2"
    m.config = config
    m._specs = V.variable_specs(config, with_lm_head=cls._with_lm_head)
    m._lib, m._handle = _FakeLib(list(m._specs)), None
    m._variables = [M.Variable(m, n, s) for n, (s, _) in m._specs.items()]
    m._pushed_trainable = {}
    m._build_layers()
    return m


def test_reference_two_stage_freezing_runs_verbatim_on_the_layer_graph():
    import wav2vec2
    cfg = wav2vec2.Wav2Vec2Config()
    model = _graph_only_model(wav2vec2.Wav2Vec2ForCTC, cfg)
    n_all = len(model.variables)
    assert n_all == 213 and [l.name for l in model.layers] == ["wav2vec2", "dropout", "lm_head"]
    backbone = model.layers[0]
    # Keras tracking order of Wav2Vec2Model's attributes: 7 conv layers, feature_projection, encoder (modeling.py:123-158)
    assert [l.name for l in backbone.layers] == [f"feature_extractor/conv_layers/{i}" for i in range(7)] + ["feature_projection", "encoder"]
    assert backbone.feature_extractor == backbone.layers[:7] and backbone.encoder is backbone.layers[-1]
    assert sum(len(l.variables) for l in model.layers) == n_all and model.count_params() == 94_396_320

    # ---- STAGE 1, src/main.py:210 verbatim ----
    model.layers[0].trainable = False
    assert [v.local_name for v in model.trainable_variables] == ["lm_head/kernel", "lm_head/bias"]
    assert model._lib.flags["masked_spec_embed"] is False and model._lib.flags["encoder/layers/11/attention/q_proj/kernel"] is False
    assert model._lib.flags["lm_head/kernel"] is True and model._lib.calls == 1      # the whole vector, in ONE native call

    # ---- STAGE 2, src/main.py:232-237 verbatim ----
    model.trainable = True
    for i in range(len(model.layers[0].layers) - 2):
        model.layers[0].layers[i].trainable = False
    frozen = [v for v in model.variables if not v.trainable]
    assert len(frozen) == 9 and all(v.local_name.startswith("feature_extractor/") for v in frozen)
    assert sum(int(np.prod(v.shape)) for v in frozen) == 4_200_448    # SURVEY a-16: the frozen conv stack
    assert sum(int(np.prod(v.shape)) for v in model.trainable_variables) == 90_195_104 + 768
    assert model._lib.flags["encoder/layers/11/attention/q_proj/kernel"] is True
    assert model._lib.flags["feature_extractor/conv_layers/3/conv/kernel"] is False
    assert not backbone.layers[0].trainable and backbone.layers[7].trainable and backbone.trainable

    # freeze_feature_extractor (modeling.py:211-214) is the same set
    other = _graph_only_model(wav2vec2.Wav2Vec2ForCTC, cfg)
    other.freeze_feature_extractor()
    assert {v.local_name for v in other.variables if not v.trainable} == {v.local_name for v in frozen}
    other.model.freeze_feature_extractor()                             # the reference's ForCTC delegates to `self.model`

    # Keras semantics: a frozen parent gates its children even if a child is switched back on
    model.layers[0].trainable = False
    model.layers[0].encoder.trainable = True
    assert not any(v.trainable for v in model.layers[0].variables)
    model.layers[0].trainable = True
    assert all(v.trainable for v in model.layers[0].encoder.variables)

    # the flat prefix API still works on top
    model.set_trainable("", False)
    model.set_trainable("lm_head/", True)
    assert [v.local_name for v in model.trainable_variables] == ["lm_head/kernel", "lm_head/bias"]
    with pytest.raises(KeyError):
        model.set_trainable("no_such_prefix/", True)
    lines = []
    model.summary(print_fn=lines.append)
    assert any("Trainable params: 24,608" in ln for ln in lines)       # 768 * 32 + 32


def test_backbone_model_layer_graph():
    import wav2vec2
    m = _graph_only_model(wav2vec2.Wav2Vec2Model, wav2vec2.RobustWav2Vec2Config())
    assert [l.name for l in m.layers][-2:] == ["feature_projection", "encoder"] and len(m.layers) == 9
    assert m.variables[0].name == "wav2vec2/masked_spec_embed:0"
    m.freeze_feature_extractor()
    # robust: conv bias + LayerNorm on every conv layer -> 4 variables per layer
    assert len([v for v in m.variables if not v.trainable]) == 28
    assert len(m.encoder.layers) == 3 + 24


def test_from_pretrained_download_failure_is_the_reference_error(monkeypatch):
    """No local directory -> the Hub is tried (reference modeling.py:57-74); without network that fails, and the failure
    is the reference's ValueError -- raised before any device is touched, so it is checkable here."""
    import wav2vec2
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    with pytest.raises(ValueError, match="Couldn't download model weights from https://huggingface.co/no-such-org/no-such-model"):
        wav2vec2.Wav2Vec2ForCTC.from_pretrained("no-such-org/no-such-model")
    assert hasattr(wav2vec2.Wav2Vec2ForCTC, "push_to_hub")


def test_bench_self_launch_builds_the_driver_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-runs itself under torch.distributed.run (one rank per GPU, rendezvous on
    127.0.0.1 at a free port) with its own arguments: the form the driver uses for N > 1, so a plain `--gpus 8` cannot fail at
    argument parsing."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1", "--precision", "bf16", "--mode", "train"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "3", "--warmup", "1", "--precision", "bf16", "--mode", "train"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under the launcher (WORLD_SIZE set) it does NOT spawn again: it goes on to the device check
    monkeypatch.setenv("WORLD_SIZE", "8")
    seen.clear()
    with pytest.raises((SystemExit, AssertionError, RuntimeError)):
        bench.main()
    assert "cmd" not in seen


def test_bench_cpu_baseline_topology_helpers(tmp_path, monkeypatch):
    """The CPU-baseline leg of bench.py sizes itself to what the host really grants: worker pin sets are disjoint, cover WHOLE cores
    (every hardware thread of a core travels with it) and stay inside this process's affinity mask; a cgroup CPU quota is read from
    cpu.max and caps the worker count (the MI355X boxes show 256 CPUs under a 16-CPU quota)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench._ranges([0, 1, 2, 128, 129, 131]) == "0-2,128-129,131" and bench._ranges([7]) == "7"
    allowed = set(os.sched_getaffinity(0))
    sets = bench._core_sets(2, 1)
    flat = [c for s in sets for c in s]
    assert len(flat) == len(set(flat)) and set(flat) <= allowed            # disjoint, inside the mask
    for s in sets:
        assert s == sorted(s) and len(s) >= 1
    assert bench._core_sets(10 ** 6, 1) == bench._core_sets(len(bench._core_sets(10 ** 6, 1)), 1)      # asks beyond the host: fewer sets, no error
    q = bench._cpu_quota()
    assert q is None or q > 0
    # cgroup v2 file format: "<quota> <period>" or "max <period>"
    real_open = open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            return real_open(tmp_path / "cpu.max", *a, **k)
        return real_open(path, *a, **k)
    (tmp_path / "cpu.max").write_text("1600000 100000\n")
    monkeypatch.setattr("builtins.open", fake_open)
    assert bench._cpu_quota() == 16.0
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert bench._cpu_quota() is None
    from oracle import w2v2_oracle as O
    (tmp_path / "cpu.max").write_text("200000 100000\n")
    assert O._usable_cpus() == min(2, len(allowed))


def test_counted_wait_kernels_have_no_scratch():
    """Build gate for the kernels whose correctness rests on hand-counted `s_waitcnt vmcnt(N)` (gemm_bf16_sw.hip: the LDS-DMA ring and the
    fp32 epilogue's early residual loads): if the register allocator spilled an asm-loaded destination, the compiler would store it
    before it has landed and the counts would no longer describe what is in flight.  hipcc cross-compiles for gfx950 on the CPU box, so the
    gate runs in the CPU suite: every instance of that file must report ScratchSize 0.  (Round-3 advisor finding; the GPU suite's
    bit-for-bit variant tests are the other half.)"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "gsoc-wav2vec2_amd", "csrc", "gemm_bf16_sw.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-c", src, "-o", os.devnull,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    names, scratch = [], []
    for line in r.stderr.splitlines():
        if "Function Name:" in line:
            names.append(line.split("Function Name:")[1].split("[")[0].strip())
        elif "ScratchSize [bytes/lane]:" in line:
            scratch.append(int(line.split("ScratchSize [bytes/lane]:")[1].split("[")[0].strip()))
    assert names and len(names) == len(scratch)
    bad = [(n, s) for n, s in zip(names, scratch) if "gemm_bf16_sw_kernel" in n and s != 0]
    assert sum("gemm_bf16_sw_kernel" in n for n in names) >= 7, names
    assert not bad, bad


def test_split_sw_kernels_have_no_scratch_and_a_consistent_schedule():
    """The same build gate for the plane-fed split GEMM (gemm_split_sw.hip: both plane formats x seven epilogue instances, all built on
    counted vmcnt waits over a ten-slot LDS ring), plus the ring / register-slot schedule simulated for every K (tools/split_sw_schedule.py:
    every item requested once and in order, no slot refilled before its reads retired behind a barrier, no item read before the counted
    wait guarantees it landed, every term multiplying the planes it claims to)."""
    import shutil
    import subprocess
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("split_sw_schedule", os.path.join(root, "tools", "split_sw_schedule.py"))
    sim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sim)
    for fmt in sim.FORMATS:
        for nk in (2, 4, 6, 8, 16, 24, 48, 96, 128):
            for half in (0, 1):
                assert sim.simulate(nk, half, fmt)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(root, "gsoc-wav2vec2_amd", "csrc", "gemm_split_sw.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-c", src, "-o", os.devnull,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-2000:]
    names, scratch = [], []
    for line in r.stderr.splitlines():
        if "Function Name:" in line:
            names.append(line.split("Function Name:")[1].split("[")[0].strip())
        elif "ScratchSize [bytes/lane]:" in line:
            scratch.append(int(line.split("ScratchSize [bytes/lane]:")[1].split("[")[0].strip()))
    assert names and len(names) == len(scratch)
    assert sum("gemm_split_sw_kernel" in n for n in names) >= 14, names
    bad = [(n, s) for n, s in zip(names, scratch) if "gemm_split_sw_kernel" in n and s != 0]
    assert not bad, bad


def test_bench_stdout_line_is_compact_and_parseable():
    """the driver keeps 8 KB of stdout and parses the last line, and round 5's 27 KB line came back `parsed: null`.
    bench.compact_line() turns the complete object into the ONE stdout line: the contract keys, `roofline` / `cpu_baseline` as numbers
    and a flat triple per side configuration, always < 4096 bytes.  Checked on the stored complete objects of round 5 (the
    headline with every side leg, and the two training lines) and on a worst case whose strings are all oversized."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test3", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for name in ("r05_bench_n1.json", "r05_bench_n1_launched.json", "r05_bench_bf16_train_n1.json", "r05_bench_large_robust_bf16_train_n1.json",
                 "r06_bench_n1.json", "r06_bench_n1_launched.json", "r06_bench_bf16_train_n1.json", "r06_bench_large_robust_bf16_train_n1.json"):
        with open(os.path.join(root, "profiles", name)) as f:
            full = json.loads(f.read().strip().splitlines()[-1])
        line = json.dumps(bench.compact_line(full))
        assert len(line) < bench.COMPACT_LIMIT == 4096, (name, len(line))
        js = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
            assert k in js, (name, k)
        assert js["value"] == full["value"] and js["ms_per_step"] == full["ms_per_step"] and js["config"]["workload"]
        assert js["roofline"]["frac"] == full["roofline"]["frac"] and js["roofline"]["bound"] == "mfma" and js["roofline"]["peak"] > 0
        assert not any(k.endswith("_note") or k in ("clock_method", "traffic_unit", "op") for k in js["roofline"])
        if "cpu_baseline" in full:
            cb = js["cpu_baseline"]
            assert cb["value"] == full["cpu_baseline"]["value"] and cb["cores"] and cb["kind"] == "port" and cb["sample"]
            assert cb["legs"]["B=1"]["median"] <= cb["legs"]["B=1"]["best"] and cb["value_median"] <= cb["value"]
        for side in ("configs2_train_bf16", "configs3_large_fwd_f32", "configs4_large_train_bf16"):
            if side in full:
                assert js[side]["ms_per_step"] == full[side]["ms_per_step"] and js[side]["frac"] == full[side]["roofline"]["frac"]
    # the line the driver itself would have read in round 6, byte for byte as bench.py printed it on the GPU box
    with open(os.path.join(root, "profiles", "r06_bench_n1_compact_line.json")) as f:
        printed = f.read().strip().splitlines()[-1]
    assert len(printed) < 4096 and json.loads(printed)["roofline"]["frac"] > 0.75 and json.loads(printed)["cpu_baseline"]["kind"] == "port"
    # worst case: every string blown up, every optional object present -> still under the limit, contract keys intact
    full["config"]["workload"] = "w" * 5000
    full["roofline"]["kernel"] = "k" * 5000
    full["cpu_baseline"] = {"value": 1.0, "unit": "audio-seconds/s", "cores": 16, "kind": "port", "sample": "s" * 5000, "cpu_model": "m" * 40,
                            "legs": {"B=1": {"audio_s_per_s_best": 2.0, "audio_s_per_s_median": 1.0}}}
    for side in ("configs2_train_bf16", "configs3_large_fwd_f32", "configs4_large_train_bf16"):
        full[side] = {"error": "e" * 5000}
    line = json.dumps(bench.compact_line(full))
    assert len(line) < 4096 and json.loads(line)["roofline"]["frac"] and json.loads(line)["cpu_baseline"]["value"] == 1.0


def test_grouped_tile_order_is_a_bijection():
    """csrc/common.h::grouped_tile (round 6: the order in which an XCD's run walks the output tiles of the wide GEMMs -- groups of gm
    tile rows, column-major inside a group).  The same integer arithmetic restated here must visit every tile exactly once for any
    tile counts and group height (the last group is shorter), and consecutive indices inside a group must stay within gm rows."""
    def grouped_tile(t, tiles_m, tiles_n, gm):
        per = gm * tiles_n
        grp = t // per
        first = grp * gm
        rows = min(gm, tiles_m - first)
        tl = t - grp * per
        tn = tl // rows
        return first + (tl - tn * rows), tn

    for tiles_m, tiles_n, gm in ((188, 16, 10), (192, 12, 12), (48, 32, 2), (11, 6, 11), (11, 6, 4), (7, 9, 3), (1, 6, 1), (17, 8, 16)):
        seen = [grouped_tile(t, tiles_m, tiles_n, gm) for t in range(tiles_m * tiles_n)]
        assert sorted(seen) == [(m, n) for m in range(tiles_m) for n in range(tiles_n)], (tiles_m, tiles_n, gm)
        for t in range(0, tiles_m * tiles_n - 64, 37):       # any 64 tiles in flight span at most two groups' rows
            rows = {m for m, _ in seen[t:t + 64]}
            assert max(rows) - min(rows) < 2 * gm + 64 // tiles_n + 1


def test_keras_weight_order_of_tf_model_h5(tmp_path):
    """`tf_model.h5` is loaded by the reference with Keras' `load_weights`, which zips the file's `weight_names`
    against the layer's weights BY POSITION -- so the order inside each HDF5 group must be the order Keras creates / lists them
    (variables.py::keras_weight_order, derived from the reference's constructors: modeling.py:158-167,227-233;
    feature_extractor.py:31-50,86-91; encoder.py:15-20,96-108,168-175,232-245; tensorflow_addons.py:23-46).  Pinned here: 213 names
    for the base CTC model (the count the reference itself prints, notebooks/wav2vec2_onnx.ipynb:125), the landmarks of the
    order, and that the file written by save_pretrained carries exactly that order and still loads (by name) to the same weights."""
    from wav2vec2 import h5lite, variables as V
    from wav2vec2.config import RobustWav2Vec2Config, Wav2Vec2Config
    cfg = Wav2Vec2Config()
    order = V.keras_weight_order(cfg, with_lm_head=True)
    assert len(order) == 213 and len(set(order)) == 213 and sorted(order) == sorted(V.variable_specs(cfg, True))
    # sub-layers in attribute-assignment order: conv stack, projection, encoder; the model's own weight after them; the head last
    assert order[0] == "feature_extractor/conv_layers/0/conv/kernel"
    assert order[1:3] == ["feature_extractor/conv_layers/0/layer_norm/gamma", "feature_extractor/conv_layers/0/layer_norm/beta"]
    assert order[3] == "feature_extractor/conv_layers/1/conv/kernel" and order[8] == "feature_extractor/conv_layers/6/conv/kernel"
    assert order[9:13] == ["feature_projection/layer_norm/gamma", "feature_projection/layer_norm/beta",
                           "feature_projection/projection/kernel", "feature_projection/projection/bias"]
    # Conv1DWithWeightNorm.build: Conv1D's kernel is replaced by weight_v (appended after the bias), then weight_g is added
    assert order[13:16] == ["encoder/pos_conv_embed/conv/bias", "encoder/pos_conv_embed/conv/weight_v", "encoder/pos_conv_embed/conv/weight_g"]
    assert order[16:18] == ["encoder/layer_norm/gamma", "encoder/layer_norm/beta"]
    l0 = [n[len("encoder/layers/0/"):] for n in order[18:34]]
    assert l0 == ["attention/q_proj/kernel", "attention/q_proj/bias", "attention/k_proj/kernel", "attention/k_proj/bias",
                  "attention/v_proj/kernel", "attention/v_proj/bias", "attention/out_proj/kernel", "attention/out_proj/bias",
                  "layer_norm/gamma", "layer_norm/beta", "feed_forward/intermediate_dense/kernel", "feed_forward/intermediate_dense/bias",
                  "feed_forward/output_dense/kernel", "feed_forward/output_dense/bias", "final_layer_norm/gamma", "final_layer_norm/beta"]
    assert order[18 + 16 * 12] == "masked_spec_embed" and order[-2:] == ["lm_head/kernel", "lm_head/bias"]
    # robust: conv bias + a LayerNorm on every conv layer -> 7 * 4 conv-stack names; 24 layers
    rorder = V.keras_weight_order(RobustWav2Vec2Config(), True)
    assert rorder[:4] == [f"feature_extractor/conv_layers/0/{s}" for s in ("conv/kernel", "conv/bias", "layer_norm/gamma", "layer_norm/beta")]
    assert len(rorder) == 1 + 28 + 4 + 3 + 2 + 24 * 16 + 2

    # the file: tiny model, CTC form -- groups [wav2vec2, dropout, lm_head], weight_names in Keras order, `:0` suffixes, full prefix
    tiny = Wav2Vec2Config(hidden_size=32, num_heads=2, num_layers=2, intermediate_size=64, filter_sizes=[16] * 7,
                          num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)
    w = V.seeded_weights(tiny, seed=3)
    path = str(tmp_path / "tf_model.h5")
    h5lite.save_keras_weights(path, V.keras_layers(tiny, w, True))
    f = h5lite.File(path)
    assert h5lite._load_names(f.root.attrs, "layer_names") == ["wav2vec2", "dropout", "lm_head"]
    names = h5lite._load_names(f.root["wav2vec2"].attrs, "weight_names")
    want = [V.tf_variable_name(n, True) for n in V.keras_weight_order(tiny, True) if not n.startswith("lm_head/")]
    assert names == want and names[-1] == "wav2vec2-ctc/wav2vec2/masked_spec_embed:0"
    assert h5lite._load_names(f.root["lm_head"].attrs, "weight_names") == ["wav2vec2-ctc/lm_head/kernel:0", "wav2vec2-ctc/lm_head/bias:0"]
    got = h5lite.load_keras_weights(path)
    assert set(got) == {V.tf_variable_name(n, True) for n in w}
    assert all(np.array_equal(got[V.tf_variable_name(n, True)], a) for n, a in w.items())

    # backbone form: one group per layer of `Wav2Vec2Model.layers`; masked_spec_embed (no layer owns it) in `top_level_model_weights`,
    # which `layer_names` does not list (Keras 2.5's by-position loader counts 9 layers with weights in file and model alike)
    wb = V.seeded_weights(tiny, seed=3, with_lm_head=False)
    pathb = str(tmp_path / "backbone.h5")
    h5lite.save_keras_weights(pathb, V.keras_layers(tiny, wb, False))
    fb = h5lite.File(pathb)
    assert h5lite._load_names(fb.root.attrs, "layer_names") == [f"feature_extractor/conv_layers/{i}" for i in range(7)] + ["feature_projection", "encoder"]
    assert h5lite._load_names(fb.root["top_level_model_weights"].attrs, "weight_names") == ["wav2vec2/masked_spec_embed:0"]
    enc = h5lite._load_names(fb.root["encoder"].attrs, "weight_names")
    assert enc[:3] == ["wav2vec2/encoder/pos_conv_embed/conv/bias:0", "wav2vec2/encoder/pos_conv_embed/conv/weight_v:0", "wav2vec2/encoder/pos_conv_embed/conv/weight_g:0"]
    gotb = h5lite.load_keras_weights(pathb)
    assert set(gotb) == {V.tf_variable_name(n, False) for n in wb} and np.array_equal(gotb["wav2vec2/masked_spec_embed:0"], wb["masked_spec_embed"])
