#!/usr/bin/env python
"""Where does the bf16-precision path leave the bf16-operand oracle?  Per-activation max error on a golden case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
from oracle import w2v2_oracle as O
import wav2vec2
name = sys.argv[1] if len(sys.argv) > 1 else "tiny_base"
g = H.golden(name); cfg = H.case_config(name); w = H.case_weights(name)
m = wav2vec2.Wav2Vec2ForCTC(cfg, input_shape=(1, 2048)); m.set_weights(w)
mask = g.get("attention_mask"); mask = None if mask is None else mask.astype(np.int32)
for prec in ("fp32", "bf16"):
    m.set_precision(prec)
    got = m(g["wave"], attention_mask=mask).numpy()
    taps = {}
    with H.oracle_operands(None if prec == "fp32" else "bf16"):
        ref = O.ctc_forward(cfg, w, g["wave"], mask, taps=taps)
    print(prec, "logits", H.max_err(got, ref))
    for k, v in taps.items():
        try:
            a = m.activation(k)
        except Exception as e:
            print("  ", k, "n/a", e); continue
        print(f"   {k:12s} max|ref| {np.abs(v).max():9.3f}  err {H.max_err(a, v):.3e}")
