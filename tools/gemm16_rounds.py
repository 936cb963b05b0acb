#!/usr/bin/env python
"""How the shadow-fed bf16 GEMM's time steps with the tile count (rounds of the chip's 512 block slots): N = 768 (three 256-column
tiles per row tile), K from the command line, M swept across one and two whole rounds; both kernels (variant 1 = 128 x 128,
2 = 128 x 256 software-pipelined).  python tools/gemm16_rounds.py [K ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd"))
import torch
from wav2vec2 import _native as N
lib = N.load(); dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
Nn = int(os.environ.get("ROUNDS_N", 768))
for K in [int(a) for a in sys.argv[1:]] or [3072, 768]:
    print(f"N = {Nn}, K = {K}: row tiles (128 x 256 tiles) -> us per launch / TF, 128 x 128 kernel | 128 x 256 sw kernel")
    for mt in (22, 64, 128, 160, 170, 171, 176, 192, 256, 320, 340, 341, 384):
        M = mt * 128
        A16 = torch.randn(M, K, device=dev).to(torch.bfloat16); B16 = (torch.randn(Nn, K, device=dev) * 0.05).to(torch.bfloat16)
        Cf = torch.empty(M, Nn, device=dev); bias = torch.randn(Nn, device=dev); R = torch.randn(M, Nn, device=dev); st = N.current_stream()
        row = []
        for v in (1, 2):
            def call():
                N.check(lib.w2v2_op_gemm_bf16_shadows(N.ptr(A16), K, 0, N.ptr(B16), N.ptr(Cf), None, Nn, 0, N.ptr(bias), N.ptr(R), M, Nn, K, 1, 0, v, st))
            t = min(timeit(call), timeit(call))
            row.append(f"{t * 1e3:7.1f} us {2.0 * M * Nn * K / t / 1e9:5.0f} TF")
        print(f"  {mt:4d} ({mt * (Nn // 256):4d})   " + "  |  ".join(row))
