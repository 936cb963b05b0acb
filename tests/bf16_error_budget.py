#!/usr/bin/env python
"""Per-stage error budget of precision mode `bf16` on the BASELINE-size fixture (CPU; test infrastructure -- it runs the
oracle, nothing here is on the product path).

    python tests/bf16_error_budget.py [--case base_sample_padded] [--out profiles/r03_bf16_error_budget.md]

The oracle runs in fp64 (its own distance to HF fp64 is ~3e-5, so everything above that is the rounding under test) with
bf16 operand rounding switched on for ONE contraction family at a time, then for the cumulative sets, against the
committed HF-PyTorch fp64 logits.  `attention core` is the build's own extra: q, k, v and the un-normalised softmax
probabilities rounded to bf16 inside the attention kernel (the reference-side definition of the mode rounds Dense /
Conv1D operands only).  The GPU-side numbers (the build vs the same fixture) are printed by
tests/test_model_gpu.py::test_bf16_precision_logits and recorded in profiles/.
"""

import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gsoc-wav2vec2_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

import helpers as H  # noqa: E402
from oracle import w2v2_oracle as O  # noqa: E402

STAGES = ["conv", "projection", "pos_conv", "qkv", "out_proj", "ffn1", "ffn2", "lm_head"]


def run(cfg, w, wave, mask, stages, attention):
    O.GEMM_OPERANDS = "bf16" if stages else None
    O.ROUND_STAGES = set(stages) if stages else None
    O.ATTENTION_OPERANDS = "bf16" if attention else None
    try:
        return O.ctc_forward(cfg, w, wave, mask, dtype=np.float64)
    finally:
        O.GEMM_OPERANDS, O.ROUND_STAGES, O.ATTENTION_OPERANDS = None, None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="base_sample_padded")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    g = H.golden(args.case)
    cfg, w = H.case_config(args.case), H.case_weights(args.case)
    mask = g.get("attention_mask")
    mask = None if mask is None else mask.astype(np.int32)
    wave, ref = g["wave"], g["logits_f64"]
    T = ref.shape[1]
    rows = []

    def measure(label, stages, attention=False):
        t0 = time.time()
        out = run(cfg, w, wave, mask, stages, attention)
        d = np.abs(out - ref)
        per_row = d.reshape(d.shape[0], -1).max(axis=1)
        where = np.unravel_index(int(d.argmax()), d.shape)
        rows.append((label, float(d.max()), float(np.sqrt((d ** 2).mean())), per_row.tolist(), int(where[1])))
        print(f"{label:44s} max {d.max():.3e}  rms {np.sqrt((d ** 2).mean()):.3e}  per row {['%.2e' % v for v in per_row]}  "
              f"worst frame {where[1]} of {T}  ({time.time() - t0:.0f} s)", flush=True)

    measure("fp64 oracle, no rounding", [])
    for st in STAGES:
        measure(f"only {st}", [st])
    measure("only the attention core (q, k, v, P)", [], attention=True)
    measure("conv stack + projection + pos_conv", ["conv", "projection", "pos_conv"])
    measure("all Dense / Conv1D (the mode's definition)", STAGES)
    measure("all Dense / Conv1D + attention core (the build)", STAGES, attention=True)
    lines = [f"# bf16 error budget -- `{args.case}` ({wave.shape[0]} x {wave.shape[1]} samples, {T} frames), CPU oracle in fp64",
             "",
             "max / rms of |logits - HF fp64 logits| with bf16 operand rounding (nearest-even) applied to ONE contraction family at a",
             "time, then cumulatively (`tests/bf16_error_budget.py`).  Logit abs-max of the fixture: "
             f"{np.abs(ref).max():.2f}.", "",
             "| rounded operands | max abs err | rms err | per row max | worst frame |", "|---|---|---|---|---|"]
    for label, mx, rms, per_row, frame in rows:
        lines.append(f"| {label} | {mx:.3e} | {rms:.3e} | {', '.join('%.2e' % v for v in per_row)} | {frame} |")
    text = "\n".join(lines) + "\n"
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main()
