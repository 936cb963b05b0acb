#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ (run in the BUILD container).

The reference package needs TensorFlow and cannot be imported here, so the
vectors come from HuggingFace-PyTorch Wav2Vec2 -- the comparator the
reference's own tests use (tests/test_wav2vec2.py:55-79 hidden states at 1e-3,
:140-157 logits at 4e-3, :217-237 CTC loss at 1e-3) -- loaded with the build's
seeded weights through the reference's checkpoint name map
(src/convert_torch_to_tf.py:12-44,110-117).  Only inputs/outputs are stored;
weights are regenerated from the seed wherever the fixtures are used.

Each case stores HF fp32 outputs and HF fp64 outputs (cast to fp32 for
storage): fp64 is the tighter "truth" the 1e-3 bar is measured against; fp32
HF-vs-fp64 HF shows the fp32 noise floor of a correct implementation.

Usage:  python tests/golden/make_golden.py [--ref /root/reference]
"""

import argparse
import os
import sys
import wave

import numpy as np
import torch
import transformers

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd"))

from wav2vec2 import variables as V                                  # noqa: E402
from wav2vec2.config import RobustWav2Vec2Config, Wav2Vec2Config     # noqa: E402

TINY = dict(hidden_size=64, num_heads=2, num_layers=2, intermediate_size=128,
            filter_sizes=[32] * 7, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)


def hf_config(c):
    return transformers.Wav2Vec2Config(
        vocab_size=c.vocab_size, hidden_size=c.hidden_size, num_hidden_layers=c.num_layers,
        num_attention_heads=c.num_heads, intermediate_size=c.intermediate_size,
        hidden_act="gelu", hidden_dropout=0.0, activation_dropout=0.0, attention_dropout=0.0,
        feat_proj_dropout=0.0, final_dropout=0.0, layerdrop=0.0, layer_norm_eps=c.layer_norm_eps,
        feat_extract_norm=c.feature_extractor_norm_type, feat_extract_activation="gelu",
        conv_dim=tuple(c.filter_sizes), conv_stride=tuple(c.strides), conv_kernel=tuple(c.kernal_sizes),
        conv_bias=c.conv_bias, num_conv_pos_embeddings=c.num_conv_pos_embeddings,
        num_conv_pos_embedding_groups=c.num_conv_pos_embedding_groups,
        do_stable_layer_norm=(c.attention_norm_type == "prenorm"), apply_spec_augment=False,
        ctc_loss_reduction="sum", ctc_zero_infinity=False, pad_token_id=c.pad_id,
        attn_implementation="eager")


def build_hf(c, seed):
    w = V.seeded_weights(c, seed=seed)
    hf = transformers.Wav2Vec2ForCTC(hf_config(c)).eval()
    sd = {k: torch.from_numpy(v) for k, v in V.to_hf_state_dict(w).items()}
    hf.load_state_dict(sd, strict=True)
    return hf


def read_wav(path):
    with wave.open(path) as f:
        assert f.getsampwidth() == 2 and f.getnchannels() == 1
        pcm = np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16)
    return pcm.astype(np.float32) / 32768.0       # tf.audio.decode_wav scaling


def normalize(x):                                  # processor.py:101-106
    x = x.astype(np.float64)
    return ((x - x.mean()) / np.sqrt(x.var() + 1e-5)).astype(np.float32)


def tap_stride(name):
    return {"conv0": 997, "conv1": 499, "conv2": 251, "conv3": 127, "conv4": 61, "conv5": 31}.get(name, 1)


def run_case(name, c, seed, x, mask, labels, out_dir, full_taps, conv_taps=True):
    print(f"[{name}] B={x.shape[0]} L={x.shape[1]} ...", flush=True)
    res = {"wave": x}
    if mask is not None:
        res["attention_mask"] = mask.astype(np.int8)
    for prec, dt in (("f32", torch.float32), ("f64", torch.float64)):
        hf = build_hf(c, seed).to(dt)
        xt = torch.from_numpy(x).to(dt)
        mt = None if mask is None else torch.from_numpy(mask.astype(np.int64))
        with torch.no_grad():
            feats = hf.wav2vec2.feature_extractor(xt).transpose(1, 2)
            out = hf.wav2vec2(xt, attention_mask=mt, output_hidden_states=True)
            logits = hf.lm_head(out.last_hidden_state)
        res[f"logits_{prec}"] = logits.float().numpy()
        if prec == "f64":
            res["conv6"] = feats.float().numpy()[:, ::1 if full_taps else 7]
            hs = out.hidden_states
            res["encoder_in"] = hs[0].float().numpy()[:, ::1 if full_taps else 13]
            res["layer0"] = hs[1].float().numpy()[:, ::1 if full_taps else 13]
            res["last_hidden"] = out.last_hidden_state.float().numpy()[:, ::1 if full_taps else 13]
            # per-layer conv taps (strided in time)
            h = xt[:, None, :]
            for i, layer in enumerate(hf.wav2vec2.feature_extractor.conv_layers if conv_taps else ()):
                with torch.no_grad():
                    h = layer(h)
                if i < 6:
                    st = 1 if full_taps else tap_stride(f"conv{i}")
                    res[f"conv{i}"] = h.transpose(1, 2).float().numpy()[:, ::st]
        if labels is not None:
            # HF CTC loss == the comparator of tests/test_wav2vec2.py:217-237.
            # All rows use the full frame count (reference losses.py:29-30).
            logp = torch.log_softmax(logits.to(torch.float64), dim=-1).transpose(0, 1)
            T = logits.shape[1]
            lab = torch.from_numpy(labels.astype(np.int64))
            lab_len = (lab != c.pad_id).sum(-1)
            flat = torch.cat([lab[b, :lab_len[b]] for b in range(lab.shape[0])])
            nll = torch.nn.functional.ctc_loss(
                logp, flat, torch.full((lab.shape[0],), T, dtype=torch.long), lab_len,
                blank=c.pad_id, reduction="none", zero_infinity=False)
            res[f"ctc_nll_{prec}"] = nll.numpy().astype(np.float64)
            res["labels"] = labels.astype(np.int32)
    e = np.abs(res["logits_f32"] - res["logits_f64"]).max()
    print(f"[{name}] logits absmax {np.abs(res['logits_f64']).max():.3f} "
          f"std {res['logits_f64'].std():.3f}  hf f32-vs-f64 {e:.2e}", flush=True)
    np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), **res)


def weight_norm_conv_case(out_dir):
    """The reference's only self-contained op test (tests/test_wav2vec2.py:
    239-282): weight-normalised grouped Conv1D == torch weight_norm(Conv1d,
    dim=2); same data recipe (np.random.seed(0) uniform (2,128,32); 16 filters,
    k=3, pad=1, groups=2), weights from the seeded generator."""
    np.random.seed(0)
    x = np.random.uniform(size=(2, 128, 32)).astype(np.float32)
    K, cg, cout, groups, pad = 3, 16, 16, 2, 1
    wv = (V.hash_uniform("wn/weight_v", K * cg * cout, 0) * 2 - 1).reshape(K, cg, cout).astype(np.float32)
    wg = (0.5 + V.hash_uniform("wn/weight_g", K, 0)).reshape(K, 1, 1).astype(np.float32)
    bias = (V.hash_uniform("wn/bias", cout, 0) - 0.5).astype(np.float32)
    conv = torch.nn.Conv1d(32, cout, K, padding=pad, groups=groups)
    conv = torch.nn.utils.weight_norm(conv, dim=2)
    conv.weight_v.data = torch.from_numpy(np.ascontiguousarray(wv.transpose(2, 1, 0)))
    conv.weight_g.data = torch.from_numpy(np.ascontiguousarray(wg.transpose(2, 1, 0)))
    conv.bias.data = torch.from_numpy(bias)
    with torch.no_grad():
        y = conv(torch.from_numpy(x).transpose(2, 1)).transpose(2, 1).numpy()
    np.savez_compressed(os.path.join(out_dir, "weight_norm_conv.npz"),
                        x=x, weight_v=wv, weight_g=wg, bias=bias, y=y)
    print("[weight_norm_conv] done", y.shape)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.manual_seed(0)
    out_dir = HERE
    only = set(filter(None, args.only.split(",")))

    def want(n):
        return not only or n in only

    # reference data fixtures (data files the reference's own tests read)
    for fn in ("sample.wav", "vocab.json"):
        src = os.path.join(args.ref, "data", fn)
        dst = os.path.join(out_dir, fn)
        if os.path.exists(src) and not os.path.exists(dst):
            with open(src, "rb") as f, open(dst, "wb") as g:
                g.write(f.read())

    pcm = read_wav(os.path.join(out_dir, "sample.wav"))
    speech = normalize(pcm)                                       # 46797 samples

    # labels recipe of tests/test_wav2vec2.py:41-42
    np.random.seed(0)
    labels = np.random.randint(1, 30, size=(2, 24))
    labels_padded = np.concatenate([labels, np.zeros((2, 8), labels.dtype)], axis=1)
    labels_padded[1, 20:] = 0                                     # ragged: 24 and 20 labels

    if want("weight_norm_conv"):
        weight_norm_conv_case(out_dir)

    if want("tiny_base"):
        c = Wav2Vec2Config(**TINY)
        x = V.hash_normal("tiny/wave", 2 * 4000, 1).reshape(2, 4000)
        run_case("tiny_base", c, 0, x, None, labels_padded[:, :8] % 31 + 0, out_dir, True)

    if want("tiny_robust"):
        c = RobustWav2Vec2Config(**TINY)
        x = V.hash_normal("tiny/wave", 2 * 4000, 2).reshape(2, 4000)
        m = np.ones((2, 4000), np.int32)
        m[0, -800:] = 0
        m[1, -37:] = 0
        x = x * m                                               # padded region is zeros (data_utils.py:62-71)
        run_case("tiny_robust", c, 0, x.astype(np.float32), m, None, out_dir, True)

    if want("base_sample_padded"):
        # BASELINE config 1/2 input convention: normalise, THEN right-pad zeros to 246000
        c = Wav2Vec2Config()
        L = 246000
        x = np.zeros((2, L), np.float32)
        x[0, :len(speech)] = speech
        x[1] = V.hash_normal("base/noise", L, 1)
        run_case("base_sample_padded", c, 0, x, None, labels_padded, out_dir, False)

    if want("base_sample_unpadded"):
        # the reference's test_inference recipe: [sample.wav ; noise] at 46797 samples
        c = Wav2Vec2Config()
        L = len(speech)
        x = np.stack([speech, V.hash_normal("base/noise", L, 2)]).astype(np.float32)
        run_case("base_sample_unpadded", c, 0, x, None, labels, out_dir, False)

    if want("robust_masked"):
        # the reference's robust recipe: last 1000 / 132 samples masked (test_wav2vec2.py:58-62)
        c = RobustWav2Vec2Config()
        L = len(speech)
        x = np.stack([speech, V.hash_normal("robust/noise", L, 3)]).astype(np.float32)
        m = np.ones((2, L), np.int32)
        m[0, -1000:] = 0
        m[1, -132:] = 0
        run_case("robust_masked", c, 0, x, m, None, out_dir, False)

    if want("robust_full_246000"):
        # BASELINE configs[3] shape: large-robust at 246000 samples, ragged mask (one full row, one row with 100000 padded
        # samples) -- the input recipe of tests/test_model_gpu.py::test_large_robust_full_length (weights seed 5)
        c = RobustWav2Vec2Config()
        L = 246000
        x = V.hash_normal("robust/full", 2 * L, 6).reshape(2, L)
        m = np.ones((2, L), np.int32)
        m[1, 146000:] = 0
        run_case("robust_full_246000", c, 5, (x * m).astype(np.float32), m, None, out_dir, False, conv_taps=False)

    if want("robust_long_480000"):
        # BASELINE configs[4] input length: 480000 samples -> T = 1499 frames, large-robust, last 70001 samples masked
        c = RobustWav2Vec2Config()
        L = 480000
        x = V.hash_normal("robust/long", L, 8).reshape(1, L)
        m = np.ones((1, L), np.int32)
        m[0, -70001:] = 0
        run_case("robust_long_480000", c, 5, (x * m).astype(np.float32), m, None, out_dir, False, conv_taps=False)

    if want("base_long_480000"):
        # the input of tests/test_model_gpu.py::test_long_form_480000 (base, seed-0 weights, no mask)
        c = Wav2Vec2Config()
        L = 480000
        x = V.hash_normal("long/wave", L, 7).reshape(1, L)
        run_case("base_long_480000", c, 0, x.astype(np.float32), None, None, out_dir, False, conv_taps=False)


if __name__ == "__main__":
    main()
