#!/usr/bin/env python
"""gemm_split_sw.hip (bf16x3 GEMM fed from pre-split planes) on the GPU: correctness against fp64 and against gemm_split.hip, the
three-plane output, ragged rows, overlapping (conv) rows with a batch, and op-level timing on the B = 32 shapes of the forward.

    python tools/split_sw_check.py [--time-only] [--reps 20]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd"))
import numpy as np, torch
from wav2vec2 import _native as N

lib = N.load(); dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
st = N.current_stream()
time_only = "--time-only" in sys.argv
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20


FMT = 1 if "--f16x2" in sys.argv else 0          # W2V2_PLANES_F16X2 | W2V2_PLANES_BF16X3
NP = 2 if FMT else 3
ACT_SCALE = 16.0
flag = torch.zeros(1, dtype=torch.int32, device=dev)


def bf16_to_f32(u16):
    return (u16.to(torch.int32) << 16).view(torch.float32)


def planes_sum(p, n):
    """(NP n) 16-bit planes -> the fp64 value they stand for"""
    if FMT:
        return (p[:n].view(torch.float16).double() + p[n:2 * n].view(torch.float16).double()) / ACT_SCALE
    return bf16_to_f32(p[:n]).double() + bf16_to_f32(p[n:2 * n]).double() + bf16_to_f32(p[2 * n:]).double()


def planes_of(t):
    """fp32 device tensor -> (NP, numel) 16-bit planes via the op; bf16x3: checked to sum back exactly, f16x2: to 2^-21 relative / 2^-28 absolute"""
    n = t.numel()
    p = torch.empty(NP * n, dtype=torch.int16, device=dev)
    N.check(lib.w2v2_op_split_planes(N.ptr(t), N.ptr(p), n, n, FMT, N.ptr(flag), st))
    torch.cuda.synchronize()
    s = planes_sum(p, n)
    if FMT:
        d = (s - t.reshape(-1).double()).abs()
        assert bool((d <= t.reshape(-1).double().abs() * 2.0 ** -21 + 2.0 ** -28).all()), "f16x2 planes off by more than 2^-21"
    else:
        assert torch.equal(s.float().double(), s) and torch.equal(s.float(), t.reshape(-1)), "planes do not sum to the fp32 value"
    return p


def images_of(tB, K, Nn):
    """-> (images, out_scale pointer tensor or None)"""
    img = torch.empty(NP * K * Nn, dtype=torch.int16, device=dev)
    ws = torch.zeros(2, dtype=torch.float32, device=dev)
    N.check(lib.w2v2_op_split_weight(N.ptr(tB), N.ptr(img), N.ptr(ws) if FMT else None, K, Nn, FMT, st))
    return img, (ws[1:] if FMT else None)


def gemm_planes(pA, planeA, lda, sA, img, C, P, planeC, ldc, sC, bias, res, M, Nn, K, nb, act):
    im, sc = img
    N.check(lib.w2v2_op_gemm_split_planes(FMT, N.ptr(pA), planeA, lda, sA, N.ptr(im), N.ptr(sc) if sc is not None else None,
                                          N.ptr(C) if C is not None else None, N.ptr(P) if P is not None else None, planeC, ldc, sC,
                                          N.ptr(bias) if bias is not None else None, N.ptr(res) if res is not None else None, M, Nn, K, nb, act, N.ptr(flag), st))


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def check(M, Nn, K, act=0, bias=True, res=False):
    A = rng.randn(M, K).astype(np.float32); B = (rng.randn(K, Nn) * 0.05).astype(np.float32)
    b = rng.randn(Nn).astype(np.float32) if bias else None
    R = rng.randn(M, Nn).astype(np.float32) if res else None
    tA, tB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    tb = torch.from_numpy(b).to(dev) if bias else None
    tR = torch.from_numpy(R).to(dev) if res else None
    pA = planes_of(tA); img = images_of(tB, K, Nn)
    n = M * K
    C = torch.empty(M, Nn, device=dev); Cold = torch.empty(M, Nn, device=dev); C32 = torch.empty(M, Nn, device=dev)
    gemm_planes(pA, n, K, 0, img, C, None, 0, Nn, 0, tb, tR, M, Nn, K, 1, act)
    N.check(lib.w2v2_op_gemm_split(N.ptr(tA), K, 0, N.ptr(tB), N.ptr(Cold), Nn, 0, N.ptr(tb) if bias else None, N.ptr(tR) if res else None, M, Nn, K, 1, act, st))
    N.check(lib.w2v2_op_gemm(N.ptr(tA), K, 0, N.ptr(tB), Nn, N.ptr(C32), Nn, 0, N.ptr(tb) if bias else None, N.ptr(tR) if res else None, M, Nn, K, 1, act, st))
    torch.cuda.synchronize()
    ref = tA.double() @ tB.double()
    if bias: ref = ref + tb.double()
    if act == 1: ref = torch.nn.functional.gelu(ref)
    if act == 2: ref = torch.nn.functional.gelu(ref, approximate="tanh")
    if res: ref = ref + tR.double()
    e = lambda x: ((x.double() - ref).abs().max().item(), (x.double() - ref).pow(2).mean().sqrt().item())
    en, eo, e3 = e(C), e(Cold), e(C32)
    line = f"M={M} N={Nn} K={K} act={act} bias={int(bias)} res={int(res)}: new max {en[0]:.2e} rms {en[1]:.2e} | old split {eo[0]:.2e} {eo[1]:.2e} | fp32 MFMA {e3[0]:.2e} {e3[1]:.2e}"
    # (the fp32 kernel splits K for small M, which shortens its summation chains: the bar is the better of it and the old split kernel;
    #  f16x2 represents its operands to 2^-22, which at K = 64 .. 256 is still above what the short fp32 sums commit: 1.5x there)
    slack = 1.5 if FMT else 1.05
    assert en[1] <= max(1.5 * e3[1], slack * eo[1]) + 1e-9 and en[0] <= max(2.0 * e3[0], 1.5 * eo[0]) + 1e-7, line
    # three-plane output == the fp32 output, bit for bit (no residual in that form)
    if not res:
        P = torch.empty(NP * M * Nn, dtype=torch.int16, device=dev)
        gemm_planes(pA, n, K, 0, img, None, P, M * Nn, Nn, 0, tb, None, M, Nn, K, 1, act)
        torch.cuda.synchronize()
        mn = M * Nn
        s = planes_sum(P, mn)
        if FMT:
            same = bool(((s - C.reshape(-1).double()).abs() <= C.reshape(-1).double().abs() * 2.0 ** -21 + 2.0 ** -28).all())
        else:
            same = torch.equal(s.float(), C.reshape(-1)) and torch.equal(s.float().double(), s)
        line += f" | planes == fp32 out: {same}"
        assert same, line
    print(line, flush=True)


def check_conv(Bn, Tin, Cin, Cout, k, s):
    """strided Conv1D as a GEMM over overlapping rows, one launch for the batch (feature_extractor.py:31-37)"""
    Tout = 1 + (Tin - k) // s
    x = rng.randn(Bn, Tin, Cin).astype(np.float32); w = (rng.randn(k * Cin, Cout) * 0.05).astype(np.float32)
    tx, tw = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    px = planes_of(tx); img = images_of(tw, k * Cin, Cout)
    out = torch.empty(Bn, Tout, Cout, device=dev)
    gemm_planes(px, tx.numel(), s * Cin, Tin * Cin, img, out, None, 0, Cout, Tout * Cout, None, None, Tout, Cout, k * Cin, Bn, 1)
    torch.cuda.synchronize()
    win = tx.double().unfold(1, k, s)                      # (B, Tout, Cin, k)
    ref = torch.nn.functional.gelu(win.permute(0, 1, 3, 2).reshape(Bn, Tout, k * Cin) @ tw.double())
    err = (out.double() - ref).abs().max().item()
    print(f"conv B={Bn} Tin={Tin} Cin={Cin} Cout={Cout} k={k} s={s}: Tout={Tout} max err {err:.2e}", flush=True)
    assert err < 2e-5


if not time_only:
    mm = torch.zeros(2, dtype=torch.int64, device=dev)
    N.check(lib.w2v2_op_check_select_forms(N.ptr(mm), st)); torch.cuda.synchronize()
    print("erf_select / tanh_select vs erff / tanhf over all 2^32 patterns: mismatches", mm.tolist(), flush=True)
    for args in [(128, 256, 64), (256, 256, 128), (300, 256, 192), (1000, 768, 768), (1, 256, 64), (129, 512, 1024),
                 (2048, 768, 3072), (4096, 2304, 768)]:
        check(*args)
    check(640, 512, 512, act=1)
    check(1000, 3072, 768, act=1)
    check(1000, 256, 256, act=2)
    check(1000, 768, 3072, act=0, res=True)
    check(777, 768, 768, act=0, bias=False, res=True)
    check(515, 512, 512, act=1, bias=False)
    check_conv(3, 1001, 64, 256, 3, 2)
    check_conv(2, 520, 128, 512, 2, 2)

# op-level timing on the forward's B = 32 shapes
print("timing (B = 32 x 246000 shapes):")
shapes = [("q|k|v", 24576, 2304, 768, 1, 0, False), ("out-proj", 24576, 768, 768, 1, 0, True), ("ffn up", 24576, 3072, 768, 1, 1, False),
          ("ffn down", 24576, 768, 3072, 1, 0, True), ("proj", 24576, 768, 512, 1, 0, False)]
tot = 0.0
for name, M, Nn, K, nb, act, res in shapes:
    tA = torch.randn(M, K, device=dev); tB = torch.randn(K, Nn, device=dev) * 0.05; tb = torch.randn(Nn, device=dev)
    tR = torch.randn(M, Nn, device=dev) if res else None
    pA = planes_of(tA); img = images_of(tB, K, Nn)
    C = torch.empty(M, Nn, device=dev)
    P = torch.empty(NP * M * Nn, dtype=torch.int16, device=dev)
    f = lambda: gemm_planes(pA, M * K, K, 0, img, C, None, 0, Nn, 0, tb, tR, M, Nn, K, 1, act)
    fp = lambda: gemm_planes(pA, M * K, K, 0, img, None, P, M * Nn, Nn, 0, tb, None, M, Nn, K, 1, act)
    t = timeit(f)
    line = f"  {name:9s} M={M} N={Nn} K={K}: fp32 out {t * 1e3:7.1f} us = {2.0 * M * Nn * K / t / 1e9:6.1f} TF"
    if not res:
        t2 = timeit(fp)
        line += f" | planes out {t2 * 1e3:7.1f} us = {2.0 * M * Nn * K / t2 / 1e9:6.1f} TF"
    print(line, flush=True)
# conv1 .. conv6 of the base extractor at B = 32 (overlapping rows, batch = samples), output planes with GELU
T = [49199, 24599, 12299, 6149, 3074, 1537, 768]
ks = [3, 3, 3, 3, 2, 2]; ss = [2, 2, 2, 2, 2, 2]
for i in range(6):
    Tin, Tout, k, s = T[i], T[i + 1], ks[i], ss[i]
    x = torch.randn(32, Tin, 512, device=dev); w = torch.randn(k * 512, 512, device=dev) * 0.03
    px = planes_of(x); img = images_of(w, k * 512, 512)
    P = torch.empty(NP * 32 * Tout * 512, dtype=torch.int16, device=dev)
    f = lambda: gemm_planes(px, x.numel(), s * 512, Tin * 512, img, None, P, 32 * Tout * 512, 512, Tout * 512, None, None, Tout, 512, k * 512, 32, 1)
    t = timeit(f)
    print(f"  conv{i + 1}     M=32x{Tout} N=512 K={k * 512}: planes out {t * 1e3:7.1f} us = {2.0 * 32 * Tout * 512 * k * 512 / t / 1e9:6.1f} TF", flush=True)
    del x, px, P
print("range flag:", flag.item())
print("split_sw_check done", "f16x2" if FMT else "bf16x3")
