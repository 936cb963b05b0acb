#!/usr/bin/env python
"""dW = X^T dY from bf16 shadows at the base model's B = 32 shapes: the 128 x 128 transposing-read kernel (variant 1) against the
128 x 256 software-pipelined kernel in its transposed form (variant 2), slab counts as the training step picks them (even for
variant 1, uneven for variant 2).  python tools/wgrad_ab.py"""
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
rows = 32 * 768
for name, Kin, Nout, S1, S2 in (("ffn1", 768, 3072, 6, 7), ("ffn2", 3072, 768, 6, 7), ("qkv", 768, 2304, 8, 9), ("out", 768, 768, 24, 28)):
    x16 = torch.randn(rows, Kin, device=dev).to(torch.bfloat16); y16 = (torch.randn(rows, Nout, device=dev) * 0.1).to(torch.bfloat16)
    out = torch.empty(32, Kin, Nout, device=dev); st = N.current_stream()
    res = []
    for variant, S in ((1, S1), (2, S1), (2, S2)):
        per = (rows // 64 // S) * 64
        def call():
            N.check(lib.w2v2_op_weight_grad_bf16(N.ptr(x16), N.ptr(y16), N.ptr(out), rows, Kin, Nout, per, S, variant, st))
        t = min(timeit(call), timeit(call))
        res.append(f"v{variant} S={S}: {t * 1e3:6.1f} us {2.0 * rows * Kin * Nout / t / 1e9:5.0f} TF")
    print(f"{name:5s} " + "   ".join(res))
