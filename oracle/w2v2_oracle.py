"""CPU oracle for the Wav2Vec2 forward / CTC path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the algorithm in the reference's
``src/wav2vec2/*.py`` (each function cites the file:line it follows).  It is
the checker the HIP path is compared with; it is NOT part of the product.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The product path
(``gsoc-wav2vec2_amd/wav2vec2``) never routes through it and fails loudly when
the HIP library is missing.

Where the arithmetic really lives: the reference calls TensorFlow
(``tensorflow==2.5``, named only in a comment at ``requirements.txt:1``), which
is absent from ``/root/reference`` and not installable here.  So this oracle
restates the published semantics of the TF ops at the reference's call sites:
``Conv1D(valid)``, ``LayerNormalization``, ``tf.nn.moments`` +
``tf.nn.batch_normalization`` (GroupNorm), exact-erf ``tf.nn.gelu``,
``tf.nn.softmax``, ``tf.matmul``, ``tf.nn.l2_normalize``, ``tf.nn.ctc_loss``.

Pinning (see tests/test_oracle_golden.py, tests/golden/make_golden.py):
  * the reference's own known-answer test for the normaliser
    (tests/test_dataloader.py:56-63, 8 samples of data/sample.wav);
  * the reference's definition of the weight-normalised grouped conv as equal
    to ``torch.nn.utils.weight_norm(Conv1d(groups), dim=2)`` at 1e-4
    (tests/test_wav2vec2.py:239-282) -- re-run against torch in the build
    container, outputs committed as a fixture;
  * HuggingFace-PyTorch Wav2Vec2 -- the comparator every model-level test of
    the reference uses at atol 1e-3 (tests/test_wav2vec2.py:55-79,140-157,
    217-237) -- run in the build container on seeded weights; logits, stage
    taps and CTC loss committed under tests/golden/.
The TF leg itself (reference-TF == HF-torch) is asserted by the reference's
tests and could not be executed here.
"""

import math
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

try:  # exact erf; scipy is present in the image
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)


# --------------------------------------------------------------------------
# elementwise / normalisation primitives
# --------------------------------------------------------------------------
_POOL = None


# Operand rounding of the dense contractions (Conv1D layers >= 1 and every Dense): None = the reference's
# fp32 arithmetic; "bf16" = both operands rounded to bfloat16 (nearest-even) before an exact product and a
# wide sum -- the restatement of W2V2_PRECISION_BF16 (include/w2v2.h), i.e. of a mixed_bfloat16 Keras policy
# on those layers.  The grouped positional conv rounds x and the weight-normalised kernel the same way (the build runs it
# as one batched GEMM in that mode); attention rounds its operands inside the kernel (not emulated here), norms and
# conv0 stay unrounded.
GEMM_OPERANDS = None


def round_bf16(x):
    """fp32 -> nearest-even bfloat16 -> back to the input dtype (values exactly representable in bf16)."""
    a = np.ascontiguousarray(x, dtype=np.float32)
    b = a.view(np.uint32).astype(np.uint64)
    b = (b + np.uint64(0x7FFF) + ((b >> np.uint64(16)) & np.uint64(1))) & np.uint64(0xFFFF0000)
    return b.astype(np.uint32).view(np.float32).reshape(a.shape).astype(np.asarray(x).dtype)


# Error-budget switches (tests/bf16_error_budget.py): with GEMM_OPERANDS == "bf16", ROUND_STAGES limits the operand rounding
# to the named contractions ("conv", "projection", "pos_conv", "qkv", "out_proj", "ffn1", "ffn2", "lm_head"; None = all of
# them, the mode the parity tests use).  ATTENTION_OPERANDS == "bf16" additionally rounds q, k, v and the softmax
# probabilities inside the attention core, which is what the build's bf16 attention kernel does (the reference-side
# definition of the mode, `_mm` only, leaves the core in the working precision).
ROUND_STAGES = None
ATTENTION_OPERANDS = None


def _mm(a, b, stage=None):
    """a @ b of one dense contraction, with the configured operand rounding."""
    if GEMM_OPERANDS == "bf16" and (ROUND_STAGES is None or stage in ROUND_STAGES):
        return round_bf16(a) @ round_bf16(b)
    return a @ b


def _usable_cpus():
    """CPUs this process can keep busy: its affinity mask (a pinned CPU-baseline worker must not start a thread per core of the
    whole host), capped by the container's cgroup CPU quota (the MI355X boxes show 256 CPUs under a 16-CPU quota: 64 runnable
    pool threads there only get each other throttled)."""
    try:
        n = max(1, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return n


def _chunked(fn, x, min_elems=1 << 20):
    """Apply an elementwise `fn` over row chunks of `x` on a thread pool (numpy / scipy ufuncs release
    the GIL).  Same arithmetic per element; only there so that the CPU baseline is not dominated by
    single-threaded ufunc loops while BLAS uses every core."""
    global _POOL
    if x.size < min_elems:
        return fn(x)
    flat = x.reshape(-1, x.shape[-1])
    n = min(64, _usable_cpus(), max(1, x.size // min_elems))
    if n <= 1:
        return fn(x)
    if _POOL is None:
        _POOL = ThreadPoolExecutor(max_workers=min(64, _usable_cpus()))
    out = np.empty_like(flat)
    bounds = np.linspace(0, flat.shape[0], n + 1).astype(int)

    def work(i):
        out[bounds[i]:bounds[i + 1]] = fn(flat[bounds[i]:bounds[i + 1]])
    list(_POOL.map(work, range(n)))
    return out.reshape(x.shape)


def _gelu_exact(x):
    return (0.5 * x * (1.0 + _erf(x * x.dtype.type(1.0 / math.sqrt(2.0))))).astype(x.dtype, copy=False)


def gelu(x, approximate=False):
    """tf.nn.gelu as called at feature_extractor.py:58, encoder.py:127,181."""
    if approximate:
        c = x.dtype.type(math.sqrt(2.0 / math.pi))
        return (0.5 * x * (1.0 + np.tanh(c * (x + 0.044715 * x ** 3)))).astype(x.dtype, copy=False)
    return _chunked(_gelu_exact, x)


def layer_norm(x, gamma, beta, eps):
    """tf.keras.layers.LayerNormalization(axis=-1): population variance over
    the last axis (feature_extractor.py:48-50,86-88; encoder.py:96-98,105-108,
    232-234)."""
    mean = x.mean(axis=-1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True)
    return ((x - mean) / np.sqrt(var + x.dtype.type(eps)) * gamma + beta).astype(x.dtype)


def group_norm_time(x, gamma, beta, eps=1e-5):
    """GroupNormalization(groups=C, axis=-1) -> the per-(sample, channel)
    statistics over TIME of tensorflow_addons.py:207-231 (tf.nn.moments over
    axis 1, population variance, then tf.nn.batch_normalization with
    gamma/beta broadcast as (1, 1, C)); called at feature_extractor.py:40-47."""
    mean = x.mean(axis=1, keepdims=True, dtype=np.float64)
    # E[(x - mean)^2] accumulated in fp64 without materialising an fp64 copy of x
    xc = x - mean.astype(x.dtype)
    var = np.einsum("btc,btc->bc", xc, xc, dtype=np.float64)[:, None, :] / x.shape[1]
    scale = (gamma / np.sqrt(var + eps)).astype(x.dtype)
    shift = (beta - (mean - mean.astype(x.dtype)) * scale).astype(x.dtype)   # residual of the fp32 centring
    return (xc * scale + shift).astype(x.dtype, copy=False)


def normalize(x):
    """Wav2Vec2Processor._normalize (processor.py:101-106): zero mean / unit
    variance along the last axis, population variance, eps 1e-5, squeeze."""
    x = np.asarray(x)
    mean = x.mean(axis=-1, keepdims=True)
    var = x.var(axis=-1, keepdims=True)
    return np.squeeze((x - mean) / np.sqrt(var + 1e-5))


# --------------------------------------------------------------------------
# convolution blocks
# --------------------------------------------------------------------------
def conv1d_valid(x, kernel, stride, bias=None):
    """tf.keras.layers.Conv1D(padding="valid") on channels-last input
    (feature_extractor.py:31-37,55).  x (B, T, Cin); kernel (K, Cin, Cout)."""
    B, T, Cin = x.shape
    K, _, Cout = kernel.shape
    T_out = 1 + (T - K) // stride
    x = np.ascontiguousarray(x)
    it = x.itemsize          # explicit strides: a size-1 axis may carry stride 0
    win = np.lib.stride_tricks.as_strided(
        x, shape=(B, T_out, K * Cin), strides=(T * Cin * it, stride * Cin * it, it), writeable=False)
    y = np.empty((B, T_out, Cout), dtype=x.dtype)
    w2 = kernel.reshape(K * Cin, Cout)
    # BLAS needs a plain (lda >= row length) matrix; the overlapping window is
    # not one, so materialise it in row chunks (numpy's fallback for exotic
    # strides is a scalar loop, ~200x slower).
    step = max(1, (32 << 20) // max(1, K * Cin * it))
    for b in range(B):
        for t0 in range(0, T_out, step):
            y[b, t0:t0 + step] = (_mm(np.ascontiguousarray(win[b, t0:t0 + step]), w2, "conv") if x.shape[2] > 1
                                   else np.ascontiguousarray(win[b, t0:t0 + step]) @ w2)   # layer 0 (C_in = 1) is not a GEMM in the build
    if bias is not None:
        y += bias
    return y


def feature_extractor(config, w, x, taps=None):
    """7 x [Conv1D -> (GroupNorm on layer 0 | LayerNorm on every layer) ->
    exact GELU] (modeling.py:188-190, feature_extractor.py:54-59)."""
    for i, (k, s) in enumerate(zip(config.kernal_sizes, config.strides)):
        base = f"feature_extractor/conv_layers/{i}"
        bias = w.get(f"{base}/conv/bias") if config.conv_bias else None
        x = conv1d_valid(x, w[f"{base}/conv/kernel"], s, bias)
        if config.feature_extractor_norm_type == "group":
            if i == 0:
                x = group_norm_time(x, w[f"{base}/layer_norm/gamma"], w[f"{base}/layer_norm/beta"], 1e-5)
        else:
            x = layer_norm(x, w[f"{base}/layer_norm/gamma"], w[f"{base}/layer_norm/beta"], 1e-5)
        x = gelu(x, config.is_gelu_approx)
        if taps is not None:
            taps[f"conv{i}"] = x
    return x


def feature_projection(config, w, x):
    """LayerNorm -> Dense(512 -> H); dropout is identity at inference
    (feature_extractor.py:92-95)."""
    x = layer_norm(x, w["feature_projection/layer_norm/gamma"],
                   w["feature_projection/layer_norm/beta"], config.layer_norm_eps)
    return _mm(x, w["feature_projection/projection/kernel"], "projection") + w["feature_projection/projection/bias"]


def weight_norm_kernel(weight_v, weight_g):
    """Conv1DWithWeightNorm._compute_kernel (tensorflow_addons.py:16-21,
    26-28): l2-normalise weight_v (K, Cin/g, Cout) over axes (1, 2) -- i.e.
    per kernel TAP -- times weight_g (K, 1, 1).  tf.nn.l2_normalize uses
    x * rsqrt(max(sum(x^2), 1e-12))."""
    ss = (weight_v.astype(np.float64) ** 2).sum(axis=(1, 2), keepdims=True)
    return (weight_v / np.sqrt(np.maximum(ss, 1e-12)) * weight_g).astype(weight_v.dtype)


def grouped_conv1d_same(x, kernel, bias, groups, padding):
    """tf.pad(padding, padding) then valid grouped Conv1D
    (tensorflow_addons.py:50-53).  x (B, T, C); kernel (K, C/groups, Cout)."""
    B, T, C = x.shape
    K, cg, Cout = kernel.shape
    og = Cout // groups
    xp = np.zeros((B, T + 2 * padding, C), dtype=x.dtype)
    xp[:, padding:padding + T] = x
    T_out = T + 2 * padding - K + 1
    y = np.empty((B, T_out, Cout), dtype=x.dtype)
    for g in range(groups):
        xg = np.ascontiguousarray(xp[:, :, g * cg:(g + 1) * cg])          # (B, Tp, cg)
        it = xg.itemsize
        Tp = xg.shape[1]
        win = np.lib.stride_tricks.as_strided(
            xg, shape=(B, T_out, K * cg), strides=(Tp * cg * it, cg * it, it), writeable=False)
        wg = kernel[:, :, g * og:(g + 1) * og].reshape(K * cg, og)
        for b in range(B):
            y[b, :, g * og:(g + 1) * og] = _mm(np.ascontiguousarray(win[b]), wg, "pos_conv")   # operand rounding as the Dense layers
    return y + bias


def pos_conv_embed(config, w, x):
    """PositionalConvEmbedding.call (encoder.py:177-181): weight-normalised
    grouped conv with explicit pad K//2 both sides, drop the last frame when K
    is even, exact GELU."""
    K = config.num_conv_pos_embeddings
    kern = weight_norm_kernel(w["encoder/pos_conv_embed/conv/weight_v"],
                              w["encoder/pos_conv_embed/conv/weight_g"])
    y = grouped_conv1d_same(x, kern, w["encoder/pos_conv_embed/conv/bias"],
                            config.num_conv_pos_embedding_groups, K // 2)
    if K % 2 == 0:
        y = y[:, :-1, :]
    return gelu(y, config.is_gelu_approx)


# --------------------------------------------------------------------------
# transformer
# --------------------------------------------------------------------------
def attention(config, w, base, x, add_mask):
    """TransformerAttention.call / get_context (encoder.py:22-47): separate
    q/k/v Dense with bias, q pre-scaled by d_h^-0.5, scores + additive mask,
    softmax(-1), P.V, merge heads, out Dense."""
    B, T, H = x.shape
    h = config.num_heads
    d = H // h

    def proj(name):
        y = _mm(x, w[f"{base}/attention/{name}/kernel"], "qkv") + w[f"{base}/attention/{name}/bias"]
        return y.reshape(B, T, h, d).transpose(0, 2, 1, 3)            # (B, h, T, d)

    q = proj("q_proj") * x.dtype.type(d ** -0.5)
    k = proj("k_proj")
    v = proj("v_proj")
    if ATTENTION_OPERANDS == "bf16":
        q, k, v = round_bf16(q), round_bf16(k), round_bf16(v)
    s = q @ k.transpose(0, 1, 3, 2)                                   # (B, h, T, T)
    if add_mask is not None:
        s = s + add_mask
    s = s - s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    denom = p.sum(axis=-1, keepdims=True)
    if ATTENTION_OPERANDS == "bf16":      # the kernel rounds the UN-normalised probabilities (operand of P V), sums them in fp32
        ctx = ((round_bf16(p) @ v) / denom).transpose(0, 2, 1, 3).reshape(B, T, H)
    else:
        ctx = ((p / denom) @ v).transpose(0, 2, 1, 3).reshape(B, T, H)
    return _mm(ctx, w[f"{base}/attention/out_proj/kernel"], "out_proj") + w[f"{base}/attention/out_proj/bias"]


def transformer_layer(config, w, i, x, add_mask):
    """TransformerLayer.call (encoder.py:111-134), inference: dropout is
    identity and StochasticDepth is a plain add (tensorflow_addons.py:386-390)."""
    base = f"encoder/layers/{i}"
    eps = config.layer_norm_eps
    pre = config.attention_norm_type == "prenorm"
    res = x
    if pre:
        x = layer_norm(x, w[f"{base}/layer_norm/gamma"], w[f"{base}/layer_norm/beta"], eps)
    x = attention(config, w, base, x, add_mask) + res
    if not pre:
        x = layer_norm(x, w[f"{base}/layer_norm/gamma"], w[f"{base}/layer_norm/beta"], eps)
    res = x
    if pre:
        x = layer_norm(x, w[f"{base}/final_layer_norm/gamma"], w[f"{base}/final_layer_norm/beta"], eps)
    x = gelu(_mm(x, w[f"{base}/feed_forward/intermediate_dense/kernel"], "ffn1")
             + w[f"{base}/feed_forward/intermediate_dense/bias"], config.is_gelu_approx)
    x = _mm(x, w[f"{base}/feed_forward/output_dense/kernel"], "ffn2") + w[f"{base}/feed_forward/output_dense/bias"]
    x = res + x
    if not pre:
        x = layer_norm(x, w[f"{base}/final_layer_norm/gamma"], w[f"{base}/final_layer_norm/beta"], eps)
    return x


def frame_lengths(config, sample_mask):
    """modeling.py:201-204: valid frames per row from a (B, L) 0/1 mask."""
    n = np.asarray(sample_mask).sum(axis=-1).astype(np.int64)
    for k, s in zip(config.kernal_sizes, config.strides):
        n = 1 + (n - k) // s
    return n


def encoder(config, w, x, frame_len=None, taps=None):
    """Wav2Vec2Encoder.call (encoder.py:251-276).  ``frame_len`` (B,) or None:
    frames >= frame_len[b] are zeroed before the positional conv and masked as
    KEYS with an additive -10000 (sequence_mask -> (1-m)*-10000 broadcast to
    (B, 1, T, T) with entry [b,0,q,k] = mask[b,k])."""
    B, T, H = x.shape
    add_mask = None
    if frame_len is not None:
        keep = (np.arange(T)[None, :] < np.asarray(frame_len)[:, None])          # (B, T)
        x = np.where(keep[:, :, None], x, x.dtype.type(0.0))
        add_mask = ((1.0 - keep.astype(x.dtype)) * x.dtype.type(-10000.0))[:, None, None, :]
    x = x + pos_conv_embed(config, w, x)
    if config.attention_norm_type == "postnorm":
        x = layer_norm(x, w["encoder/layer_norm/gamma"], w["encoder/layer_norm/beta"], config.layer_norm_eps)
    if taps is not None:
        taps["encoder_in"] = x
    for i in range(config.num_layers):
        x = transformer_layer(config, w, i, x, add_mask)
        if taps is not None and (i == 0 or i == config.num_layers - 1):
            taps[f"layer{i}"] = x
    if config.attention_norm_type == "prenorm":
        x = layer_norm(x, w["encoder/layer_norm/gamma"], w["encoder/layer_norm/beta"], config.layer_norm_eps)
    return x


def model_forward(config, w, wave, attention_mask=None, taps=None, dtype=np.float32):
    """Wav2Vec2Model.call at inference (modeling.py:169-209).  wave (B, L);
    attention_mask (B, L) of 0/1 or None.  Returns (B, T, H)."""
    w = {k: np.asarray(v, dtype=dtype) for k, v in w.items()}
    x = np.asarray(wave, dtype=dtype)[:, :, None]
    x = feature_extractor(config, w, x, taps)
    x = feature_projection(config, w, x)
    if taps is not None:
        taps["projection"] = x
    frame_len = frame_lengths(config, attention_mask) if attention_mask is not None else None
    return encoder(config, w, x, frame_len, taps)


def ctc_forward(config, w, wave, attention_mask=None, taps=None, dtype=np.float32):
    """Wav2Vec2ForCTC.call at inference (modeling.py:239-255): backbone ->
    (dropout: identity) -> lm_head Dense(H -> vocab).  Returns logits (B,T,V)."""
    h = model_forward(config, w, wave, attention_mask, taps, dtype)
    return _mm(h, np.asarray(w["lm_head/kernel"], dtype=dtype), "lm_head") + np.asarray(w["lm_head/bias"], dtype=dtype)


# --------------------------------------------------------------------------
# CTC loss
# --------------------------------------------------------------------------
def _logsumexp2(a, b):
    m = np.maximum(a, b)
    m_safe = np.where(np.isfinite(m), m, 0.0)
    return np.where(np.isfinite(m), m_safe + np.log(np.exp(a - m_safe) + np.exp(b - m_safe)), -np.inf)


def ctc_nll(logits, labels, label_length, logit_length, blank=0):
    """Per-sample CTC negative log-likelihood -- what tf.nn.ctc_loss returns for
    logits_time_major=False (losses.py:35-43): log-softmax over the vocab, the
    standard alpha recursion over the blank-interleaved label string."""
    logits = np.asarray(logits, dtype=np.float64)
    B, T, V = logits.shape
    m = logits.max(axis=-1, keepdims=True)
    logp = logits - m - np.log(np.exp(logits - m).sum(axis=-1, keepdims=True))
    out = np.zeros(B, dtype=np.float64)
    for b in range(B):
        U = int(label_length[b])
        Tb = int(logit_length[b])
        ext = np.full(2 * U + 1, blank, dtype=np.int64)
        ext[1::2] = np.asarray(labels[b][:U], dtype=np.int64)
        S = 2 * U + 1
        alpha = np.full(S, -np.inf)
        alpha[0] = logp[b, 0, blank]
        if S > 1:
            alpha[1] = logp[b, 0, ext[1]]
        can_skip = np.zeros(S, dtype=bool)
        can_skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
        for t in range(1, Tb):
            a1 = np.concatenate(([-np.inf], alpha[:-1]))
            a2 = np.concatenate(([-np.inf, -np.inf], alpha[:-2]))
            a2 = np.where(can_skip, a2, -np.inf)
            alpha = _logsumexp2(_logsumexp2(alpha, a1), a2) + logp[b, t, ext]
        tot = alpha[S - 1] if S == 1 else _logsumexp2(alpha[S - 1], alpha[S - 2])
        out[b] = -tot
    return out


def ctc_loss(config, labels, logits, model_input_shape, division_factor=1):
    """CTCLoss.call + Keras Reduction.SUM (losses.py:4-56): every row uses the
    FULL frame count derived from the static model input length (not the real
    audio length); label_length = count of labels != pad_id; blank = pad_id."""
    labels = np.asarray(labels)
    B = labels.shape[0]
    T = model_input_shape[1]
    for k, s in zip(config.kernal_sizes, config.strides):
        T = 1 + (T - k) // s
    logit_length = np.full(B, T, dtype=np.int64)
    label_length = (labels != config.pad_id).sum(axis=-1)
    nll = ctc_nll(logits, labels, label_length, logit_length, blank=config.pad_id)
    return float((nll / division_factor).sum()), nll


# --------------------------------------------------------------------------
# host post-processing (greedy CTC decode)
# --------------------------------------------------------------------------
def greedy_ids(logits):
    return np.asarray(logits).argmax(axis=-1)
