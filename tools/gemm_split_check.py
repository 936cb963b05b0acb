#!/usr/bin/env python
"""Accuracy and speed of w2v2_op_gemm_split (fp32 as 3 x bf16, six MFMA products) next to the native fp32 MFMA GEMM.
Error is measured against an fp64 product of the same fp32 operands."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd"))
import numpy as np, torch
from wav2vec2 import _native as N

lib = N.load(); dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
st = N.current_stream()

def run(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for (M, Nn, K) in [(300, 256, 128), (1000, 768, 96), (4096, 768, 3072), (24576, 768, 768), (24576, 3072, 768), (24576, 768, 3072), (24576, 2304, 768)]:
    A = rng.randn(M, K).astype(np.float32); B = (rng.randn(K, Nn) * 0.05).astype(np.float32)
    bias = rng.randn(Nn).astype(np.float32)
    tA, tB, tb = [torch.from_numpy(x).to(dev) for x in (A, B, bias)]
    C0 = torch.empty(M, Nn, device=dev); C1 = torch.empty(M, Nn, device=dev)
    f32 = lambda: N.check(lib.w2v2_op_gemm(N.ptr(tA), K, 0, N.ptr(tB), Nn, N.ptr(C0), Nn, 0, N.ptr(tb), None, M, Nn, K, 1, 0, st))
    spl = lambda: N.check(lib.w2v2_op_gemm_split(N.ptr(tA), K, 0, N.ptr(tB), N.ptr(C1), Nn, 0, N.ptr(tb), None, M, Nn, K, 1, 0, st))
    f32(); spl(); torch.cuda.synchronize()
    rows = slice(0, min(M, 2048))
    ref = A[rows].astype(np.float64) @ B.astype(np.float64) + bias
    e0 = np.abs(C0.cpu().numpy()[rows] - ref); e1 = np.abs(C1.cpu().numpy()[rows] - ref)
    scale = np.abs(ref).mean()
    t0 = run(f32, 5)
    print(f"M={M} N={Nn} K={K}: fp32 MFMA max {e0.max():.3e} rms {np.sqrt((e0**2).mean()):.3e} | split max {e1.max():.3e} rms {np.sqrt((e1**2).mean()):.3e}"
          f" | mean|C| {scale:.3f} | fp32 {t0:.3f} ms = {2.0*M*Nn*K/t0/1e9:.1f} TF", flush=True)
