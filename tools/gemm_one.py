#!/usr/bin/env python
"""Run ONE GEMM shape repeatedly (for rocprofv3 --pmc passes). python tools/gemm_one.py qkv 20"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd"))
import torch
from wav2vec2 import _native as N
name = sys.argv[1]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10; bf16 = len(sys.argv) > 3 and sys.argv[3] == "bf16"; split = len(sys.argv) > 3 and sys.argv[3] == "split"; planes = len(sys.argv) > 3 and sys.argv[3] == "planes"
B = 32; BT = B * 768
S = {"conv1": (24599, 512, 1536, 1024, 49199 * 512, B, 1), "qkv": (BT, 2304, 768, 768, 0, 1, 0),
     "ffn1": (BT, 3072, 768, 768, 0, 1, 1), "ffn2": (BT, 768, 3072, 3072, 0, 1, 0), "out": (BT, 768, 768, 768, 0, 1, 0)}
M, Nn, K, lda, sA, nb, act = S[name]
lib = N.load(); dev = torch.device("cuda:0")
a_elems = (nb - 1) * sA + (M - 1) * lda + K if sA else M * lda
A = torch.randn(a_elems, device=dev); Bm = torch.randn(K, Nn, device=dev) * 0.05
C = torch.empty(nb * M * Nn, device=dev); bias = torch.randn(Nn, device=dev)
st = N.current_stream()
if planes:      # the plane-fed bf16x3 kernel (gemm_split_sw.hip): planes of A and weight images built once, outside the loop
    pA = torch.empty(3 * a_elems, dtype=torch.int16, device=dev); img = torch.empty(3 * K * Nn, dtype=torch.int16, device=dev)
    FMT = 1 if os.environ.get("SPLIT_FMT") == "f16x2" else 0
    ws = torch.zeros(2, device=dev)
    N.check(lib.w2v2_op_split_planes(N.ptr(A), N.ptr(pA), a_elems, a_elems - a_elems % 4, FMT, None, st))
    N.check(lib.w2v2_op_split_weight(N.ptr(Bm), N.ptr(img), N.ptr(ws) if FMT else None, K, Nn, FMT, st))
    P = torch.empty(3 * nb * M * Nn, dtype=torch.int16, device=dev) if act or name == "conv1" else None
for _ in range(iters):
    if planes:
        N.check(lib.w2v2_op_gemm_split_planes(FMT, N.ptr(pA), a_elems, lda, sA, N.ptr(img), N.ptr(ws[1:]) if FMT else None, None if P is not None else N.ptr(C),
                                              N.ptr(P) if P is not None else None, nb * M * Nn, Nn, M * Nn, N.ptr(bias), None, M, Nn, K, nb, act, None, st))
        continue
    if split:
        N.check(lib.w2v2_op_gemm_split(N.ptr(A), lda, sA, N.ptr(Bm), N.ptr(C), Nn, M * Nn, N.ptr(bias), None, M, Nn, K, nb, act, st))
        continue
    N.check((lib.w2v2_op_gemm_bf16 if bf16 else lib.w2v2_op_gemm)(N.ptr(A), lda, sA, N.ptr(Bm), Nn, N.ptr(C), Nn, M * Nn, N.ptr(bias), None, M, Nn, K, nb, act, st))
torch.cuda.synchronize()
print("done", name)
