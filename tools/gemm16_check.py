#!/usr/bin/env python
"""Quick numerical check of w2v2_op_gemm_bf16 against a numpy emulation (bf16-rounded operands, fp64 sum)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd"))
import numpy as np, torch
from wav2vec2 import _native as N

def rbf16(x):
    b = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000
    return b.astype(np.uint32).view(np.float32)

lib = N.load(); dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
for (M, Nn, K) in [(300, 130, 128), (257, 32, 192), (128, 128, 64), (1000, 770, 100), (65, 3, 7)]:
    A = rng.randn(M, K).astype(np.float32); B = (rng.randn(K, Nn) * 0.1).astype(np.float32)
    bias = rng.randn(Nn).astype(np.float32); R = rng.randn(M, Nn).astype(np.float32)
    ref = rbf16(A).astype(np.float64) @ rbf16(B).astype(np.float64) + bias + R
    tA, tB, tb, tR = [torch.from_numpy(x).to(dev) for x in (A, B, bias, R)]
    C = torch.empty(M, Nn, device=dev)
    N.check(lib.w2v2_op_gemm_bf16(N.ptr(tA), K, 0, N.ptr(tB), Nn, N.ptr(C), Nn, 0, N.ptr(tb), N.ptr(tR), M, Nn, K, 1, 0, N.current_stream()))
    torch.cuda.synchronize()
    err = np.abs(C.cpu().numpy() - ref).max()
    exact = np.abs(A.astype(np.float64) @ B.astype(np.float64) + bias + R - ref).max()
    print(f"M={M} N={Nn} K={K}: max err vs bf16-operand emulation {err:.3e}   (bf16 rounding itself moves the result by {exact:.3e})")
