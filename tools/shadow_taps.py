#!/usr/bin/env python
"""bf16 mode with vs without shadows: per-activation max difference (must be 0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gsoc-wav2vec2_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import helpers as H
import wav2vec2
name = sys.argv[1] if len(sys.argv) > 1 else "base_sample_unpadded"
g = H.golden(name); cfg = H.case_config(name); w = H.case_weights(name)
m = wav2vec2.Wav2Vec2ForCTC(cfg, input_shape=(1, 2048)); m.set_weights(w); m.set_precision("bf16")
mask = g.get("attention_mask"); mask = None if mask is None else mask.astype(np.int32)
taps = [f"conv{i}" for i in range(7)] + ["projection", "encoder_in"] + [f"layer{i}" for i in range(cfg.num_layers)]
res = {}
for flag in ("0", "1"):
    m.set_option("bf16_shadows", flag == "1"); m.set_option("keep_activations", True)
    out = m(g["wave"], attention_mask=mask).numpy()
    res[flag] = {k: m.activation(k) for k in taps}
    res[flag]["logits"] = out
for k in taps + ["logits"]:
    print(f"{k:12s} {np.abs(res['0'][k] - res['1'][k]).max():.3e}   max|x| {np.abs(res['0'][k]).max():.3f}")
