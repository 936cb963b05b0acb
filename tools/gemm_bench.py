#!/usr/bin/env python
"""Time the GEMM shapes of the wav2vec2-base forward (B=32 x 246000) one by one through w2v2_op_gemm.
Usage: python tools/gemm_bench.py [--iters 10]"""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd"))
import torch
from wav2vec2 import _native as N

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=10); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--bf16", action="store_true")
args = ap.parse_args()
lib = N.load(); dev = torch.device("cuda:0"); B = args.batch
T = [49199, 24599, 12299, 6149, 3074, 1537, 768]
shapes = []   # name, M, N, K, lda, strideA, nbatch, act, bias, res
for i, (k, s) in enumerate([(3, 2), (3, 2), (3, 2), (3, 2), (2, 2), (2, 2)], start=1):
    shapes.append((f"conv{i}", T[i], 512, k * 512, s * 512, T[i - 1] * 512, B, 1, False, False))
BT = B * 768
shapes += [("proj", BT, 768, 512, 512, 0, 1, 0, True, False), ("qkv", BT, 2304, 768, 768, 0, 1, 0, True, False),
           ("out", BT, 768, 768, 768, 0, 1, 0, True, True), ("ffn1", BT, 3072, 768, 768, 0, 1, 1, True, False),
           ("ffn2", BT, 768, 3072, 3072, 0, 1, 0, True, True), ("lm_head", BT, 32, 768, 768, 0, 1, 0, True, False)]
res = {}
for name, M, Nn, K, lda, sA, nb, act, ub, ur in shapes:
    a_elems = (nb - 1) * sA + (M - 1) * lda + K if sA else M * lda
    A = torch.randn(a_elems, device=dev); Bm = torch.randn(K, Nn, device=dev) * 0.05
    C = torch.empty(nb * M * Nn, device=dev); bias = torch.randn(Nn, device=dev); R = torch.randn(nb * M * Nn, device=dev) if ur else None
    st = N.current_stream()
    def run():
        N.check((lib.w2v2_op_gemm_bf16 if args.bf16 else lib.w2v2_op_gemm)(N.ptr(A), lda, sA, N.ptr(Bm), Nn, N.ptr(C), Nn, M * Nn, N.ptr(bias) if ub else None,
                                 N.ptr(R), M, Nn, K, nb, act, st))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    tf = 2.0 * M * Nn * K * nb / ms / 1e9
    res[name] = dict(ms=round(ms, 4), tflops=round(tf, 1))
    print(f"{name:8s} M={M:6d} N={Nn:5d} K={K:5d} batch={nb:3d}  {ms:8.3f} ms  {tf:6.1f} TF")
print(json.dumps(res))
