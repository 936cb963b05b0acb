#!/usr/bin/env python
"""Model-level check of the plane-fed precision modes (bf16x3 with producer-written planes, f16x2) against the HF fp64 fixtures at a
batch large enough for every GEMM to take the plane kernel: logit / CTC error, row independence, range flag, time per forward.

    python tools/planes_model_check.py [base_sample_padded|robust_full_246000] [copies]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import helpers as H
import wav2vec2

name = sys.argv[1] if len(sys.argv) > 1 else "base_sample_padded"
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 8
from wav2vec2 import variables as V
g = H.golden(name); cfg = H.case_config(name)
w = V.seeded_weights(cfg, seed=5) if name == "robust_full_246000" else H.case_weights(name)      # (that fixture's weights: tests/test_model_gpu.py)
m = wav2vec2.Wav2Vec2ForCTC(cfg, input_shape=(1, 2048)); m.set_weights(w)
wave = np.concatenate([g["wave"]] * copies, 0)
mask = g.get("attention_mask")
mask = None if mask is None else np.concatenate([mask.astype(np.int32)] * copies, 0)
ref = g["logits_f64"].astype(np.float64)
nb = g["wave"].shape[0]
loss_fn = wav2vec2.CTCLoss(cfg, g["wave"].shape, division_factor=1) if "labels" in g else None


def run(tag, prec, planes=True, keep=False):
    m.set_precision(prec)
    m.set_option("split_planes", planes)
    m.set_option("keep_activations", keep)
    out = m(wave, attention_mask=mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = m(wave, attention_mask=mask)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    lg = out.numpy()
    err = np.abs(lg[:nb].astype(np.float64) - ref).max()
    same_rows = all(np.array_equal(lg[:nb], lg[k * nb:(k + 1) * nb]) for k in range(1, copies))
    line = f"{tag:36s} max|logits - HF fp64| {err:.2e}  copies bit-equal {same_rows}  {ms:7.2f} ms / forward (B = {wave.shape[0]})"
    if loss_fn is not None:
        nll = loss_fn.per_sample(g["labels"], out[:nb]).cpu().numpy()
        line += f"  nll err {np.abs(nll - g['ctc_nll_f64']).max():.2e}"
    if prec == "f16x2":
        line += f"  range flag {m.range_overflow()}"
    print(line, flush=True)
    return lg


a = run("fp32", "fp32")
b = run("bf16x3, planes", "bf16x3")
b2 = run("bf16x3, planes (again: reproducible)", "bf16x3")
print("   bitwise reproducible:", np.array_equal(b, b2))
c = run("bf16x3, planes off (round-4 path)", "bf16x3", planes=False)
print("   planes vs no planes max diff:", float(np.abs(b - c).max()))
d = run("bf16x3, planes, keep_activations", "bf16x3", keep=True)
print("   keep_activations changes nothing:", np.array_equal(b, d))
for tap in ("conv0", "conv3", "encoder_in", "layer0"):
    e = H.max_err(H.tap_view(tap, m.activation(tap)[:nb], False), g[tap]) if tap in g else float("nan")
    print(f"   tap {tap}: {e:.2e}")
f = run("f16x2", "f16x2")
f2 = run("f16x2 (again)", "f16x2")
print("   bitwise reproducible:", np.array_equal(f, f2))
print("   f16x2 vs fp32 path max diff:", float(np.abs(f - a).max()), " bf16x3 vs fp32:", float(np.abs(b - a).max()))
m.set_precision("fp32")
print("planes_model_check done")
