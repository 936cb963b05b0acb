#!/usr/bin/env python
"""Time arbitrary GEMM shapes: python tools/gemm_shape.py M,N,K[,act] ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd"))
import torch
from wav2vec2 import _native as N
lib = N.load(); dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    v = [int(x) for x in spec.split(",")]
    M, Nn, K = v[:3]; act = v[3] if len(v) > 3 else 0
    A = torch.randn(M, K, device=dev); B = torch.randn(K, Nn, device=dev) * 0.05; C = torch.empty(M, Nn, device=dev)
    bias = torch.randn(Nn, device=dev); st = N.current_stream()
    run = lambda: N.check(lib.w2v2_op_gemm(N.ptr(A), K, 0, N.ptr(B), Nn, N.ptr(C), Nn, 0, N.ptr(bias), None, M, Nn, K, 1, act, st))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"M={M} N={Nn} K={K} act={act}: {ms:.3f} ms  {2.0*M*Nn*K/ms/1e9:.1f} TF")
