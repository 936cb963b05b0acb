#!/usr/bin/env python
"""A/B of the two shadow-fed bf16 GEMM kernels through the product op w2v2_op_gemm_bf16_shadows: variant 1 (gemm_bf16_kernel,
128 x 128 tiles) against variant 2 (gemm_bf16_sw_kernel, 128 x 256 software-pipelined, two 4-wave blocks per CU): bitwise
comparison of the fp32 and bf16 outputs on ragged and model shapes, then interleaved timing on the B = 32 model shapes.

    python tools/gemm16_ab.py [--time-only]            (W2V2_NATIVE_LIB selects the tools-only build for knob sweeps)"""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd"))
import ctypes as C
import torch
from wav2vec2 import _native as N
ap = argparse.ArgumentParser(); ap.add_argument("--time-only", action="store_true"); ap.add_argument("--prio", default="1")
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
lib = N.load()
P, I32, I64 = C.c_void_p, C.c_int32, C.c_int64
dev = torch.device("cuda:0")

def make(M, Nn, K, lda, sA, nb, f32o, b16o, res, seed=0):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    a_elems = (nb - 1) * sA + (M - 1) * lda + K if sA else M * lda
    A16 = torch.randn(a_elems, device=dev, generator=g).to(torch.bfloat16)
    B16 = (torch.randn(Nn, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(Nn, device=dev, generator=g)
    R = torch.randn(nb * M * Nn, device=dev, generator=g) if res else None
    return A16, B16, bias, R

def run(pp, t, M, Nn, K, lda, sA, nb, act, f32o, b16o):
    A16, B16, bias, R = t
    Cf = torch.full((nb * M * Nn,), float("nan"), device=dev) if f32o else None
    Ch = torch.zeros(nb * M * Nn, device=dev, dtype=torch.bfloat16) if b16o else None
    N.check(lib.w2v2_op_gemm_bf16_shadows(N.ptr(A16), lda, sA, N.ptr(B16), N.ptr(Cf), N.ptr(Ch), Nn, M * Nn, N.ptr(bias), N.ptr(R), M, Nn, K, nb, act, pp, N.current_stream()))
    torch.cuda.synchronize()
    return Cf, Ch

if not args.time_only:
    # (M, N, K, lda, strideA, batch, act, fp32 out, bf16 out, residual)
    cases = [(256, 256, 256, 256, 0, 1, 0, True, True, False), (512, 768, 768, 768, 0, 1, 1, True, True, True), (300, 260, 384, 384, 0, 1, 0, True, False, True),
             (1000, 520, 512, 512, 0, 1, 1, False, True, False), (257, 1, 256, 256, 0, 1, 0, True, True, False), (2459, 512, 1536, 1024, 4919 * 512, 3, 1, True, True, False),
             (768 * 4, 2304, 768, 768, 0, 1, 0, False, True, False), (768 * 2, 768, 3072, 3072, 0, 1, 0, True, False, True),
             (130, 256, 192, 192, 0, 1, 0, True, True, False), (1000, 512, 448, 448, 0, 2, 1, False, True, False)]
    for d in ("-",):
        for c in cases:
            M, Nn, K, lda, sA, nb, act, f32o, b16o, res = c
            t = make(M, Nn, K, lda, sA, nb, f32o, b16o, res)
            ok = True
            for rep in range(3):
                c0, h0 = run(1, t, M, Nn, K, lda, sA, nb, act, f32o, b16o)
                c1, h1 = run(2 if Nn % 256 == 0 and K % 64 == 0 and K >= 192 else 0, t, M, Nn, K, lda, sA, nb, act, f32o, b16o)
                same = (c0 is None or torch.equal(c0, c1)) and (h0 is None or torch.equal(h0.view(torch.int16), h1.view(torch.int16)))
                fin = (c1 is None or bool(torch.isfinite(c1).all()))
                ok &= same and fin
                if not same:
                    dc = float((c0 - c1).abs().max()) if c0 is not None else float((h0.float() - h1.float()).abs().max())
                    print(f"   MISMATCH {c}: max diff {dc:.3e}")
                    break
            print(f"M={M} N={Nn} K={K} batch={nb} act={act} f32={f32o} bf16={b16o} res={res}: {'identical bits x3' if ok else 'DIFFERENT'}")

B = 32; BT = B * 768
SHAPES = {"qkv": (BT, 2304, 768, 768, 0, 1, 0, False, True, False), "out": (BT, 768, 768, 768, 0, 1, 0, True, False, True),
          "ffn1": (BT, 3072, 768, 768, 0, 1, 1, False, True, False), "ffn2": (BT, 768, 3072, 3072, 0, 1, 0, True, False, True),
          "qkv_dx": (BT, 768, 2304, 2304, 0, 1, 0, True, True, False), "ffn1_dx": (BT, 768, 3072, 3072, 0, 1, 0, True, True, False),
          "ffn2_dx": (BT, 3072, 768, 768, 0, 1, 0, True, False, False),
          "conv1": (24599, 512, 1536, 1024, 49199 * 512, B, 1, False, True, False), "conv3": (6149, 512, 1536, 1024, 12299 * 512, B, 1, False, True, False),
          "conv5": (1537, 512, 1024, 1024, 3074 * 512, B, 1, False, True, False), "proj": (BT, 768, 512, 512, 0, 1, 0, True, False, False)}
def timeit(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
print(f"{'shape':8s} {'128x128':>16s} {'128x256 sw':>18s} {'auto (row split)':>18s}   auto == 128x128 bits")
for name, (M, Nn, K, lda, sA, nb, act, f32o, b16o, res) in SHAPES.items():
    A16, B16, bias, R = make(M, Nn, K, lda, sA, nb, f32o, b16o, res)
    Cf = torch.empty(nb * M * Nn, device=dev) if f32o else None
    Ch = torch.empty(nb * M * Nn, device=dev, dtype=torch.bfloat16) if b16o else None
    st = N.current_stream()
    variant = [1]
    def call():
        N.check(lib.w2v2_op_gemm_bf16_shadows(N.ptr(A16), lda, sA, N.ptr(B16), N.ptr(Cf), N.ptr(Ch), Nn, M * Nn, N.ptr(bias), N.ptr(R), M, Nn, K, nb, act, variant[0], st))
    fl = 2.0 * M * Nn * K * nb
    ts = {}
    for rep in range(2):
        for v in (1, 2, 0):
            variant[0] = v
            t = timeit(call, args.iters)
            ts[v] = min(ts.get(v, 1e9), t)
    outs = {}
    for v in (1, 0):
        variant[0] = v
        if Cf is not None: Cf.fill_(float("nan"))
        call(); torch.cuda.synchronize()
        outs[v] = (Cf.clone() if Cf is not None else None, Ch.clone() if Ch is not None else None)
    same = all(a is None or torch.equal(a.view(torch.int32 if a.dtype == torch.float32 else torch.int16), b.view(torch.int32 if b.dtype == torch.float32 else torch.int16))
               for a, b in zip(outs[1], outs[0]))
    print(f"{name:8s} " + " ".join(f"{ts[v] * 1e3:7.1f} us {fl / ts[v] / 1e9:5.0f} TF" for v in (1, 2, 0)) + f"   {'identical' if same else 'DIFFERENT'}")
