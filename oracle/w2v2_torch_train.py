"""Training-step oracle (TEST INFRASTRUCTURE ONLY): the reference's training-mode forward written in
plain torch (fp64, CPU) so that autograd supplies reference gradients.

Follows the same reference lines as oracle/w2v2_oracle.py plus the train-only branches:
Dropout at feature_extractor.py:95, encoder.py:42-44,118,128,270, modeling.py:253; spec-augment
``tf.where(mask, masked_spec_embed, x)`` (spec_augment.py:119-127); StochasticDepth train branch
``shortcut + b * residual`` (tensorflow_addons.py:381-384); CTC loss / division_factor, SUM
(losses.py:6,45).  Dropout masks are NOT drawn here: they are the build's counter-based hash
(wav2vec2/variables.py::dropout_keep), so the HIP path and this oracle see the same masks.
Only tests may import this module.
"""

import math

import numpy as np
import torch

from wav2vec2 import variables as V


# Operand rounding of the Dense contractions, mirroring oracle/w2v2_oracle.py::GEMM_OPERANDS.  "bf16": both
# operands of every Dense are rounded to bfloat16 in the forward with a straight-through gradient, so autograd
# yields dX = dY . bf16(W)^T and dW = bf16(X)^T . dY; the build additionally rounds dY in its backward GEMMs
# (W2V2_PRECISION_BF16), which is why bf16 gradients are compared at a bf16-sized tolerance.  The same mode stores the FFN
# pre-activation as bf16 (rounded below, straight-through) and, in the backward, the gradient of the FFN hidden activation.
GEMM_OPERANDS = None


def _r(t):
    if GEMM_OPERANDS == "bf16":
        return t + (t.to(torch.float32).to(torch.bfloat16).to(t.dtype) - t).detach()
    return t


def _mm(a, b):
    return _r(a) @ _r(b)


def _drop(x, p, seed, stream):
    if p <= 0.0:
        return x
    keep = torch.from_numpy(V.dropout_keep(seed, stream, x.numel(), p).reshape(tuple(x.shape)))
    return torch.where(keep, x / (1.0 - p), torch.zeros_like(x))


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _ln(x, g, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def conv_stack(config, w, wave):
    """Frozen feature extractor (no dropout inside): numpy oracle, no gradient needed."""
    from oracle import w2v2_oracle as O
    wn = {k: v.detach().numpy() for k, v in w.items() if k.startswith("feature_extractor/")}
    x = np.asarray(wave, dtype=np.float64)[:, :, None]
    prev, O.GEMM_OPERANDS = O.GEMM_OPERANDS, GEMM_OPERANDS
    try:
        return torch.from_numpy(O.feature_extractor(config, wn, x))
    finally:
        O.GEMM_OPERANDS = prev


def train_forward(config, w, wave, attention_mask=None, p=0.0, seed=0, spec_mask=None, sd_keep=None,
                  checkpoint_layers=False):
    """w: {local_name: torch.float64 tensor (requires_grad for trainables)}.  Returns logits (B, T, V)."""
    pre = config.attention_norm_type == "prenorm"
    c = config
    eps = c.layer_norm_eps
    x0 = conv_stack(c, w, wave)
    B, T, _ = x0.shape
    H, h = c.hidden_size, c.num_heads
    d = H // h
    x = _ln(x0, w["feature_projection/layer_norm/gamma"], w["feature_projection/layer_norm/beta"], eps)
    x = _mm(x, w["feature_projection/projection/kernel"]) + w["feature_projection/projection/bias"]
    x = _drop(x, p, seed, V.DS_FEATURE_PROJECTION)
    if spec_mask is not None:
        m = torch.from_numpy(np.asarray(spec_mask).astype(bool))[:, :, None]
        x = torch.where(m, w["masked_spec_embed"][None, None, :].expand_as(x), x)
    add_mask = None
    if attention_mask is not None:
        from oracle import w2v2_oracle as O
        flen = O.frame_lengths(c, attention_mask)
        keep = torch.from_numpy(np.arange(T)[None, :] < np.asarray(flen)[:, None])
        x = torch.where(keep[:, :, None], x, torch.zeros_like(x))
        add_mask = ((~keep).to(x.dtype) * -10000.0)[:, None, None, :]
    # positional conv: weight-norm per tap, pad K/2, grouped, drop last for even K, GELU, residual
    K, G = c.num_conv_pos_embeddings, c.num_conv_pos_embedding_groups
    wv, wg = w["encoder/pos_conv_embed/conv/weight_v"], w["encoder/pos_conv_embed/conv/weight_g"]
    kern = wv / torch.sqrt(torch.clamp((wv ** 2).sum((1, 2), keepdim=True), min=1e-12)) * wg      # (K, cg, H)
    y = torch.nn.functional.conv1d(_r(x).transpose(1, 2), _r(kern).permute(2, 1, 0), w["encoder/pos_conv_embed/conv/bias"],
                                   padding=K // 2, groups=G).transpose(1, 2)
    if K % 2 == 0:
        y = y[:, :-1]
    x = x + _gelu(y)
    if not pre:                                                      # encoder.py:267-268
        x = _ln(x, w["encoder/layer_norm/gamma"], w["encoder/layer_norm/beta"], eps)
    x = _drop(x, p, seed, V.DS_ENCODER_IN)
    def layer(x, i):
        b = f"encoder/layers/{i}"

        def proj(name, t):
            return (_mm(t, w[f"{b}/attention/{name}/kernel"]) + w[f"{b}/attention/{name}/bias"]).reshape(B, T, h, d).transpose(1, 2)

        res = x
        a_in = _ln(x, w[f"{b}/layer_norm/gamma"], w[f"{b}/layer_norm/beta"], eps) if pre else x   # encoder.py:114-115
        q = proj("q_proj", a_in) * d ** -0.5
        k, v = proj("k_proj", a_in), proj("v_proj", a_in)
        s = q @ k.transpose(-1, -2)
        if add_mask is not None:
            s = s + add_mask
        pr = torch.softmax(s, -1)
        if p > 0.0:     # attention probabilities: index space with an even row stride (V.attention_keep)
            keep = torch.from_numpy(np.ascontiguousarray(V.attention_keep(seed, V.layer_stream(i, 0), pr.numel() // T, T, p)).reshape(tuple(pr.shape)))
            pr = torch.where(keep, pr / (1.0 - p), torch.zeros_like(pr))
        ctx = (pr @ v).transpose(1, 2).reshape(B, T, H)
        o = _mm(ctx, w[f"{b}/attention/out_proj/kernel"]) + w[f"{b}/attention/out_proj/bias"]
        x = _drop(o, p, seed, V.layer_stream(i, 1)) + res
        if not pre:
            x = _ln(x, w[f"{b}/layer_norm/gamma"], w[f"{b}/layer_norm/beta"], eps)
        keep_l = 1.0 if sd_keep is None else float(sd_keep[i])
        if keep_l != 0.0:
            f_in = _ln(x, w[f"{b}/final_layer_norm/gamma"], w[f"{b}/final_layer_norm/beta"], eps) if pre else x
            u = _mm(f_in, w[f"{b}/feed_forward/intermediate_dense/kernel"]) + w[f"{b}/feed_forward/intermediate_dense/bias"]
            # (W2V2_PRECISION_BF16 keeps this pre-activation as bf16 in the training step, as a mixed_bfloat16 Dense hands it to its
            #  activation: GELU and, through the straight-through rounding, GELU' see the rounded value)
            g = _drop(_gelu(_r(u)), p, seed, V.layer_stream(i, 2))
            f = _mm(g, w[f"{b}/feed_forward/output_dense/kernel"]) + w[f"{b}/feed_forward/output_dense/bias"]
            x = x + keep_l * f
        if not pre:
            x = _ln(x, w[f"{b}/final_layer_norm/gamma"], w[f"{b}/final_layer_norm/beta"], eps)
        return x

    for i in range(c.num_layers):
        if checkpoint_layers:
            # long inputs (T = 1499, 24 layers): keep one layer's T x T tensors alive at a time; every mask is a pure
            # function of (seed, site, index), so the recomputation in the backward reproduces the forward exactly
            from torch.utils.checkpoint import checkpoint
            x = checkpoint(layer, x, i, use_reentrant=False)
        else:
            x = layer(x, i)
    if pre:                                                          # encoder.py:274-275
        x = _ln(x, w["encoder/layer_norm/gamma"], w["encoder/layer_norm/beta"], eps)
    x = _drop(x, p, seed, V.DS_HEAD)
    return _mm(x, w["lm_head/kernel"]) + w["lm_head/bias"]


def ctc_loss_sum(config, logits, labels, division_factor=1.0):
    """CTCLoss.call + Reduction.SUM with the reference's full-T logit length (losses.py:29-45)."""
    B, T, _ = logits.shape
    lab = torch.from_numpy(np.asarray(labels).astype(np.int64))
    lab_len = (lab != config.pad_id).sum(-1)
    flat = torch.cat([lab[b, :lab_len[b]] for b in range(B)])
    logp = torch.log_softmax(logits, -1).transpose(0, 1)
    nll = torch.nn.functional.ctc_loss(logp, flat, torch.full((B,), T, dtype=torch.long), lab_len,
                                       blank=config.pad_id, reduction="none", zero_infinity=False)
    return (nll / division_factor).sum(), nll


def loss_and_grads(config, weights, wave, labels, attention_mask=None, p=0.0, seed=0, spec_mask=None,
                   sd_keep=None, division_factor=1.0, trainable=None, checkpoint_layers=False):
    """Reference loss, per-sample nll, logits and {name: gradient} for the trainable variables."""
    w = {}
    for k, v in weights.items():
        t = torch.from_numpy(np.asarray(v, dtype=np.float64).copy())
        frozen = k.startswith("feature_extractor/") or (trainable is not None and not trainable(k))
        t.requires_grad_(not frozen)
        w[k] = t
    logits = train_forward(config, w, wave, attention_mask, p, seed, spec_mask, sd_keep, checkpoint_layers)
    loss, nll = ctc_loss_sum(config, logits, labels, division_factor)
    loss.backward()
    grads = {k: (t.grad.numpy() if t.grad is not None else None) for k, t in w.items() if t.requires_grad}
    return float(loss.detach()), nll.detach().numpy(), logits.detach().numpy(), grads


def adam_reference(p, g, m, v, lr, b1, b2, eps, step):
    """Keras Adam (non-amsgrad) as tf.keras.optimizers.Adam applies it."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    return p - lr_t * m / (np.sqrt(v) + eps), m, v
