"""Variable inventory, checkpoint naming and the seeded weight generator.

Variable names and layouts are the reference's TF checkpoint layout, which is
only written down in ``src/convert_torch_to_tf.py:12-44,92-123`` (HF key ->
TF name; 2-D kernels ``(in, out)``; 3-D conv kernels ``(K, C_in, C_out)``;
``weight_g`` ``(K, 1, 1)``).  Names used across the C-ABI are those TF names
without the model prefix (``wav2vec2-ctc/wav2vec2/`` or ``wav2vec2/``) and
without the ``:0`` suffix.

The generator is a counter-based integer hash (splitmix64) so that the GPU
box regenerates bit-identical weights from a seed: no weights, and nothing of
the reference, has to travel.  It depends on integer arithmetic only, not on
any library's distribution streams.
"""

from collections import OrderedDict

import numpy as np

PREFIX_WITH_HEAD = "wav2vec2-ctc/"       # convert_torch_to_tf.py:24
PREFIX_WITHOUT_HEAD = "wav2vec2/"        # convert_torch_to_tf.py:30
SUFFIX = ":0"                            # convert_torch_to_tf.py:12


def variable_specs(config, with_lm_head=True):
    """Ordered ``{local_name: (shape, kind)}`` for one model.

    ``kind`` in {"kernel", "bias", "gamma", "beta", "embed", "weight_g",
    "weight_v"} drives the seeded generator's scale.  Local names are relative
    to the backbone (``Wav2Vec2Model``); the LM head is ``lm_head/...``.
    """
    c = config
    H, F = c.hidden_size, c.intermediate_size
    specs = OrderedDict()
    specs["masked_spec_embed"] = ((H,), "embed")                   # modeling.py:161-167
    c_in = 1
    for i, (c_out, k) in enumerate(zip(c.filter_sizes, c.kernal_sizes)):
        base = f"feature_extractor/conv_layers/{i}"
        specs[f"{base}/conv/kernel"] = ((k, c_in, c_out), "kernel")
        if c.conv_bias:
            specs[f"{base}/conv/bias"] = ((c_out,), "bias")
        has_norm = (c.feature_extractor_norm_type == "layer") or i == 0   # feature_extractor.py:39-50
        if has_norm:
            specs[f"{base}/layer_norm/gamma"] = ((c_out,), "gamma")
            specs[f"{base}/layer_norm/beta"] = ((c_out,), "beta")
        c_in = c_out
    specs["feature_projection/layer_norm/gamma"] = ((c_in,), "gamma")
    specs["feature_projection/layer_norm/beta"] = ((c_in,), "beta")
    specs["feature_projection/projection/kernel"] = ((c_in, H), "kernel")
    specs["feature_projection/projection/bias"] = ((H,), "bias")
    K, G = c.num_conv_pos_embeddings, c.num_conv_pos_embedding_groups
    specs["encoder/pos_conv_embed/conv/bias"] = ((H,), "bias")
    specs["encoder/pos_conv_embed/conv/weight_g"] = ((K, 1, 1), "weight_g")
    specs["encoder/pos_conv_embed/conv/weight_v"] = ((K, H // G, H), "weight_v")
    specs["encoder/layer_norm/gamma"] = ((H,), "gamma")
    specs["encoder/layer_norm/beta"] = ((H,), "beta")
    for i in range(c.num_layers):
        base = f"encoder/layers/{i}"
        for p in ("q_proj", "k_proj", "v_proj", "out_proj"):
            specs[f"{base}/attention/{p}/kernel"] = ((H, H), "kernel")
            specs[f"{base}/attention/{p}/bias"] = ((H,), "bias")
        specs[f"{base}/layer_norm/gamma"] = ((H,), "gamma")
        specs[f"{base}/layer_norm/beta"] = ((H,), "beta")
        specs[f"{base}/feed_forward/intermediate_dense/kernel"] = ((H, F), "kernel")
        specs[f"{base}/feed_forward/intermediate_dense/bias"] = ((F,), "bias")
        specs[f"{base}/feed_forward/output_dense/kernel"] = ((F, H), "kernel")
        specs[f"{base}/feed_forward/output_dense/bias"] = ((H,), "bias")
        specs[f"{base}/final_layer_norm/gamma"] = ((H,), "gamma")
        specs[f"{base}/final_layer_norm/beta"] = ((H,), "beta")
    if with_lm_head:
        specs["lm_head/kernel"] = ((H, c.vocab_size), "kernel")
        specs["lm_head/bias"] = ((c.vocab_size,), "bias")
    return specs


def keras_weight_order(config, with_lm_head=True):
    """Local variable names in the order Keras (TF 2.5, the reference's version) lists the model's weights -- the order of the
    ``weight_names`` attribute of each HDF5 group, which is what ``load_weights`` of a ``tf_model.h5`` zips against (by position
    inside a top-level layer, NOT by name).  Derived from the reference's constructors by Keras' rules, no TF run (none exists
    here):

    * a ``Layer`` lists its OWN weights (``add_weight`` order) and then its tracked sub-layers in ATTRIBUTE-ASSIGNMENT order
      (``__init__`` order, not call order); list attributes are tracked in list order.  Assignment order in the reference:
      ``FeatureExtractorLayer``: conv_layer, layer_norm (feature_extractor.py:31-50) -- ``Conv1D`` kernel then bias,
      ``GroupNormalization`` / ``LayerNormalization`` gamma then beta; ``FeatureProjection``: layer_norm, projection (:86-91);
      ``Wav2Vec2Encoder``: pos_conv_embed, layer_norm, dropout, layers[i] (encoder.py:232-250); ``TransformerLayer``: attention
      (q, k, v, projection: :15-18), dropout, layer_norm, intermediate, attn_output, final_layer_norm, stochastic_depth (:96-108).
      This is the order ``variable_specs`` already uses, with two exceptions:
    * ``Conv1DWithWeightNorm.build`` (tensorflow_addons.py:23-36) first runs ``Conv1D.build`` (kernel, bias), then REPLACES the
      attribute ``kernel`` by a new ``tf.Variable(name="weight_v")`` -- Keras' ``__setattr__`` drops the replaced variable from
      the layer's weight list and appends the new one -- and only then adds ``weight_g``: bias, weight_v, weight_g.
    * ``Wav2Vec2Model`` is a ``tf.keras.Model``: ``Model.trainable_weights`` walks the tracked sub-objects FIRST and appends the
      model's own ``_trainable_weights`` (``masked_spec_embed``, modeling.py:161-167) LAST.  [from the Keras 2.4 / 2.5 sources as
      remembered: unverified here; loading in this package is by name, so this affects only files carried back to the reference]

    The count pins the derivation to the reference's own output: 213 names for the base CTC model
    (notebooks/wav2vec2_onnx.ipynb:125, "Total number of loaded variables: 213" -- a 214th, the orphaned ``kernel`` of the
    positional conv, would appear if the replaced variable stayed listed)."""
    specs = variable_specs(config, with_lm_head)
    body = [n for n in specs if n != "masked_spec_embed" and not n.startswith("lm_head/")]
    pc = "encoder/pos_conv_embed/conv/"
    i = body.index(pc + "bias")
    assert body[i:i + 3] == [pc + "bias", pc + "weight_g", pc + "weight_v"]
    body[i:i + 3] = [pc + "bias", pc + "weight_v", pc + "weight_g"]
    order = body + ["masked_spec_embed"]
    if with_lm_head:
        order += ["lm_head/kernel", "lm_head/bias"]
    assert sorted(order) == sorted(specs)
    return order


def keras_layers(config, weights, with_lm_head):
    """``[(group, [(TF variable name, array)])]`` of the Keras HDF5 weight file of one model, groups in ``model.layers`` order and
    weights inside a group in ``keras_weight_order``.  ``Wav2Vec2ForCTC.layers`` = [wav2vec2, dropout, lm_head]
    (modeling.py:227-229); ``Wav2Vec2Model.layers`` = the 7 conv layers, feature_projection, encoder -- its own weight
    ``masked_spec_embed`` belongs to no layer: Keras 2.5 leaves it out of the file, Keras >= 2.6 stores such weights in the group
    ``top_level_model_weights`` (not listed in ``layer_names``); it is written there, and read back from there by this package."""
    order = keras_weight_order(config, with_lm_head)
    named = lambda n: (tf_variable_name(n, with_lm_head), weights[n])
    if with_lm_head:
        return [("wav2vec2", [named(n) for n in order if not n.startswith("lm_head/")]), ("dropout", []),
                ("lm_head", [named(n) for n in order if n.startswith("lm_head/")])]
    groups, seq = {}, []
    for n in order:
        parts = n.split("/")
        g = "/".join(parts[:3]) if n.startswith("feature_extractor/conv_layers/") else parts[0] if len(parts) > 1 else TOP_LEVEL_GROUP
        if g not in groups:
            groups[g] = []
            seq.append(g)
        groups[g].append(named(n))
    return [(g, groups[g]) for g in seq]


TOP_LEVEL_GROUP = "top_level_model_weights"      # Keras >= 2.6: weights of the model itself (no layer owns them)


def tf_variable_name(local_name, with_lm_head=True):
    """Full TF variable name as the reference's converter spells it
    (convert_torch_to_tf.py:24-35,38-44)."""
    if with_lm_head:
        if local_name.startswith("lm_head/"):
            return PREFIX_WITH_HEAD + local_name + SUFFIX
        return PREFIX_WITH_HEAD + "wav2vec2/" + local_name + SUFFIX
    return PREFIX_WITHOUT_HEAD + local_name + SUFFIX


def local_name_from_tf(tf_name):
    n = tf_name[:-len(SUFFIX)] if tf_name.endswith(SUFFIX) else tf_name
    for p in (PREFIX_WITH_HEAD + "wav2vec2/", PREFIX_WITH_HEAD, PREFIX_WITHOUT_HEAD):
        if n.startswith(p):
            return n[len(p):]
    return n


# --------------------------------------------------------------------------
# HF-PyTorch <-> TF-layout key mapping (inverse of convert_torch_to_tf.py).
# --------------------------------------------------------------------------
def hf_key_for(local_name, with_lm_head=True):
    """HuggingFace ``state_dict`` key for a local variable name, and a tag
    naming the layout transform between the two (``"T2"`` 2-D transpose,
    ``"T3"`` 3-D axis reversal, ``""`` none) -- convert_torch_to_tf.py:110-117."""
    n = local_name
    tag = ""
    if n.endswith("/kernel"):
        n = n[:-len("/kernel")] + "/weight"
        tag = "T3" if "/conv/" in local_name else "T2"
    elif n.endswith("/gamma"):
        n = n[:-len("/gamma")] + "/weight"
    elif n.endswith("/beta"):
        n = n[:-len("/beta")] + "/bias"
    elif n.endswith("weight_g") or n.endswith("weight_v"):
        tag = "T3"
    key = n.replace("/", ".")
    if with_lm_head and not key.startswith("lm_head."):
        key = "wav2vec2." + key
    return key, tag


def to_hf_state_dict(weights, with_lm_head=True, new_weight_norm_names=True):
    """TF-layout weights -> HF state_dict (numpy arrays).  HF >= 4.3x names
    the pos-conv weight-norm params ``parametrizations.weight.original0/1``
    (g / v) instead of ``weight_g/weight_v``."""
    out = {}
    for name, arr in weights.items():
        key, tag = hf_key_for(name, with_lm_head)
        a = np.asarray(arr)
        if tag == "T2":
            a = a.T
        elif tag == "T3":
            a = np.transpose(a, (2, 1, 0))
        if new_weight_norm_names:
            key = key.replace("conv.weight_g", "conv.parametrizations.weight.original0")
            key = key.replace("conv.weight_v", "conv.parametrizations.weight.original1")
        out[key] = np.ascontiguousarray(a)
    return out


def from_hf_state_dict(state_dict, config, with_lm_head=True):
    """HF state_dict (anything with ``.numpy()`` or array-like values) ->
    TF-layout weights; the forward direction of convert_torch_to_tf.py."""
    sd = {}
    for k, v in state_dict.items():
        k = k.replace("conv.parametrizations.weight.original0", "conv.weight_g")
        k = k.replace("conv.parametrizations.weight.original1", "conv.weight_v")
        sd[k] = v
    out = OrderedDict()
    for name, (shape, _) in variable_specs(config, with_lm_head).items():
        key, tag = hf_key_for(name, with_lm_head)
        if key not in sd:
            raise KeyError(f"state_dict has no `{key}` (for `{name}`)")
        v = sd[key]
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        if tag == "T2":
            a = a.T
        elif tag == "T3":
            a = np.transpose(a, (2, 1, 0))
        a = np.ascontiguousarray(a, dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"`{name}`: expected {shape}, got {a.shape}")
        out[name] = a
    return out


# --------------------------------------------------------------------------
# Seeded, counter-based generator.
# --------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x):
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(tag: str, n: int, seed: int = 0):
    """``n`` float32 values uniform in [0, 1) with 24-bit resolution, a pure
    function of (tag, seed, index)."""
    base = np.uint64((_fnv1a64(tag) ^ ((seed * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF)))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + base
    z = _splitmix64(ctr)
    return ((z >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / (1 << 24))


def hash_normal(tag: str, n: int, seed: int = 0):
    """Approximately N(0,1) float32 noise (sum of 4 uniforms, variance-matched):
    the build's own reproducible stand-in for the ``tf.random.normal`` rows the
    reference's tests use (tests/test_wav2vec2.py:37-38), which need TF."""
    u = hash_uniform(tag, 4 * n, seed).reshape(4, n).astype(np.float64)
    return ((u.sum(0) - 2.0) * np.sqrt(3.0)).astype(np.float32)


def seeded_weights(config, seed=0, with_lm_head=True):
    """Random but healthy weights in TF layout: variance-preserving uniform
    kernels, gammas near 1, small betas/biases."""
    out = OrderedDict()
    for name, (shape, kind) in variable_specs(config, with_lm_head).items():
        n = int(np.prod(shape))
        u = hash_uniform(name, n, seed).astype(np.float64) * 2.0 - 1.0   # [-1, 1)
        if kind in ("kernel", "weight_v"):
            fan_in = int(np.prod(shape[:-1]))
            gain = 1.0
            if "feature_extractor" in name:
                gain = 1.6          # conv -> GELU chain: keep the signal O(1)
            if name.startswith("lm_head"):
                gain = 2.0
            a = u * gain * np.sqrt(3.0 / fan_in)
        elif kind == "weight_g":
            # per-tap norms of a kernel drawn like weight_v, jittered by +-20 %
            K = shape[0]
            a = (1.0 + 0.2 * u) * np.sqrt(1.0 / K)
        elif kind == "gamma":
            a = 1.0 + 0.1 * u
        elif kind in ("beta", "bias"):
            a = 0.05 * u
        elif kind == "embed":
            a = 0.5 * (u + 1.0)     # Keras "uniform" initializer flavour
        else:
            raise ValueError(kind)
        out[name] = a.astype(np.float32).reshape(shape)
    return out


# --------------------------------------------------------------------------
# Dropout mask hash -- the host-side twin of csrc/train.h::dropout_hash.
# --------------------------------------------------------------------------
DS_FEATURE_PROJECTION, DS_ENCODER_IN, DS_HEAD, DS_LAYER_BASE = 1, 2, 3, 16


def layer_stream(layer, site):
    """site: 0 attention probabilities, 1 attention output, 2 FFN intermediate."""
    return DS_LAYER_BASE + 4 * layer + site


def dropout_hash(seed, stream, n, start=0):
    """16-bit hash value per element index, identical to csrc/train.h (integer arithmetic only).  The eight elements of an OCT
    (index >> 3) share one mixer round x of ((oct mod 2^32) * 0x9E3779B1) xor a per-(seed, stream) key; the 64-bit products
    x * 0x846ca68b (elements 0-3) and x * 0xC2B2AE35 (elements 4-7) give two words each -- lo ^ (lo >> 16) and hi + (lo << 16) --
    whose low half belongs to the even element and high half to the odd one."""
    key0 = np.array([(int(seed) ^ ((int(stream) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF],
                    dtype=np.uint64)
    key = np.uint32(int(_splitmix64(key0)[0]) & 0xFFFFFFFF)
    idx = np.arange(start, start + n, dtype=np.uint64)
    lo = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    u16, u15 = np.uint32(16), np.uint32(15)
    with np.errstate(over="ignore"):
        x = ((lo >> np.uint32(3)) * np.uint32(0x9E3779B1)) ^ key
        x = x ^ (x >> u16)
        x = x * np.uint32(0x7FEB352D)
        x = x ^ (x >> u15)
        mult = np.where((lo & np.uint32(4)) != 0, np.uint64(0xC2B2AE35), np.uint64(0x846CA68B))
        prod = x.astype(np.uint64) * mult
        plo, phi = (prod & np.uint64(0xFFFFFFFF)).astype(np.uint32), (prod >> np.uint64(32)).astype(np.uint32)
        w = np.where((lo & np.uint32(2)) != 0, phi + (plo << u16), plo ^ (plo >> u16))
    return np.where((lo & np.uint32(1)) != 0, w >> u16, w & np.uint32(0xFFFF)).astype(np.uint32)


def dropout_keep(seed, stream, n, p):
    """Boolean keep mask: element kept iff its 16-bit hash value, read as a signed number, >= floor(p * 2^16) - 2^15 -- i.e.
    (value ^ 0x8000) >= floor(p * 2^16) -- then scaled by 1 / (1 - p).  (csrc/train.h: the signed form lets the device take both
    decisions of a hash word with two packed 16-bit instructions.)"""
    thr = np.uint32(min(int(float(np.float32(p)) * 65536.0), 0xFFFF))
    return (dropout_hash(seed, stream, n) ^ np.uint32(0x8000)) >= thr


def attention_keep(seed, stream, rows, T, p):
    """Keep mask of the attention probabilities, (rows, T) with rows = B * heads * T queries: the element index space has the row
    stride T rounded up to a multiple of 16 and, inside a row, key k at column k with bits 2 and 3 exchanged
    (csrc/train.h::attention_drop_stride / attention_drop_col: a hash oct is then what one lane of the forward kernel holds)."""
    T2 = (T + 15) & ~15
    k = np.arange(T)
    col = (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1)
    return dropout_keep(seed, stream, rows * T2, p).reshape(rows, T2)[:, col]
