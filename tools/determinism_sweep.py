#!/usr/bin/env python
"""Run-to-run bitwise reproducibility sweep: attention at several shapes / block sizes (fp32, bf16), whole forward of base and
robust in the three precision modes.   python tools/determinism_sweep.py"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsoc-wav2vec2_amd")]
import numpy as np, torch
import wav2vec2
from wav2vec2 import _native as N, variables as V
lib = N.load(); dev = torch.device("cuda:0")
# op-level attention (inference) at several shapes, masked and not, fp32 / bf16
rs = np.random.RandomState(0)
for prec in (0, 1):
    N.check(lib.w2v2_op_set_precision(prec))
    for (B, T, heads) in ((1, 97, 2), (2, 150, 2), (4, 768, 12), (16, 768, 12), (2, 1499, 16), (32, 768, 12)):
        Hh = heads * 64
        qkv = torch.from_numpy(rs.randn(B, T, 3 * Hh).astype(np.float32)).to(dev)
        flen = torch.from_numpy(np.maximum(1, T - rs.randint(0, T // 2, size=B)).astype(np.int32)).to(dev)
        first = None
        for it in range(8):
            ctx = torch.zeros((B, T, Hh), device=dev)
            N.check(lib.w2v2_op_attention(N.ptr(qkv), N.ptr(flen), N.ptr(ctx), B, T, Hh, heads, N.current_stream()))
            torch.cuda.synchronize()
            o = ctx.cpu().numpy()
            if first is None: first = o
            elif not np.array_equal(o, first): print("NONDETERMINISTIC attention", prec, B, T, heads, int((o != first).sum())); break
        else:
            print("ok attention", prec, B, T, heads, "finite", bool(np.isfinite(first).all()))
N.check(lib.w2v2_op_set_precision(0))
# model level
for name, cfg in (("base", wav2vec2.Wav2Vec2Config()), ("robust", wav2vec2.RobustWav2Vec2Config())):
    for prec in ("fp32", "bf16", "bf16x3"):
        B, L = (4, 246000) if name == "base" else (2, 160000)
        m = wav2vec2.Wav2Vec2ForCTC(cfg, input_shape=(B, L)); m.set_precision(prec)
        x = torch.randn((B, L), device=dev)
        mask = torch.ones((B, L), device=dev, dtype=torch.int32) if cfg.is_robust else None
        if mask is not None: mask[1, -30000:] = 0
        outs = [m(x, attention_mask=mask).cpu().numpy().copy() for _ in range(5)]
        print("model", name, prec, "deterministic" if all(np.array_equal(o, outs[0]) for o in outs) else "NONDETERMINISTIC")
        del m
