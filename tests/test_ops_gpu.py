"""Per-kernel parity: every HIP kernel family, through the C ABI, against the CPU oracle
primitive of the same op on the same seeded inputs.  fp32; tolerances written per test."""

import ctypes as C

import numpy as np
import pytest

import helpers as H
from oracle import w2v2_oracle as O
from wav2vec2 import _native as N
from wav2vec2 import variables as V

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    lib = N.load()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    return lib, torch, dev


def rnd(tag, shape, scale=1.0):
    n = int(np.prod(shape))
    return ((V.hash_uniform(tag, n, 11) * 2 - 1) * scale).reshape(shape).astype(np.float32)


_KEEP = []


@pytest.fixture(autouse=True)
def _keepalive():
    """Raw pointers are handed to the C ABI: every device tensor made by dev_t must outlive the call
    (a temporary would be freed, and its block reused, before the kernel runs)."""
    _KEEP.clear()
    yield
    _KEEP.clear()


def dev_t(torch, dev, a):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    _KEEP.append(t)
    return t


def stream():
    return N.current_stream()


# ---------------------------------------------------------------- GEMM ----------
@pytest.mark.parametrize("M,N_,K,act,use_bias,use_res", [
    (128, 128, 32, 0, False, False),
    (300, 200, 96, 1, True, True),       # ragged M and N tiles
    (257, 32, 768, 0, True, False),      # lm_head-like narrow N
    (64, 2304, 64, 0, True, False),      # packed qkv-like wide N
    (130, 100, 50, 2, True, True),       # K not a multiple of 32 -> guarded path
    (77, 30, 19, 1, True, False),        # nothing aligned
    (1100, 896, 64, 0, True, True),      # 9 x 7 tiles of 128 x 128: the grouped tile order (common.h::grouped_tile), ragged last group
    (2300, 1664, 48, 1, True, False),    # 9 x 13 tiles of 256 x 128 / 18 x 13 of 128 x 128: groups of several rows, ragged rows and columns
])
def test_gemm_matches_numpy(env, M, N_, K, act, use_bias, use_res):
    lib, torch, dev = env
    A, B = rnd("A", (M, K)), rnd("B", (K, N_), 0.2)
    bias = rnd("bias", (N_,)) if use_bias else None
    res = rnd("res", (M, N_)) if use_res else None
    ref = A.astype(np.float64) @ B.astype(np.float64)
    if use_bias:
        ref = ref + bias
    if act:
        ref = O.gelu(ref, approximate=(act == 2))
    if use_res:
        ref = ref + res
    tA, tB = dev_t(torch, dev, A), dev_t(torch, dev, B)
    tb = dev_t(torch, dev, bias) if use_bias else None
    tr = dev_t(torch, dev, res) if use_res else None
    out = torch.full((M, N_), float("nan"), device=dev)
    N.check(lib.w2v2_op_gemm(N.ptr(tA), K, 0, N.ptr(tB), N_, N.ptr(out), N_, 0, N.ptr(tb), N.ptr(tr),
                             M, N_, K, 1, act, stream()))
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert H.max_err(got, ref) < 2e-5 * max(1.0, np.abs(ref).max())


def test_gemm_transpose_detecting(env):
    """A = I with an ASYMMETRIC B: a swapped C row/col map cannot pass."""
    lib, torch, dev = env
    n = 160
    A = np.eye(n, dtype=np.float32)
    B = (np.arange(n)[:, None] * 1000 + np.arange(n)[None, :]).astype(np.float32)
    out = torch.empty((n, n), device=dev)
    N.check(lib.w2v2_op_gemm(N.ptr(dev_t(torch, dev, A)), n, 0, N.ptr(dev_t(torch, dev, B)), n, N.ptr(out), n, 0,
                             None, None, n, n, n, 1, 0, stream()))
    assert np.array_equal(out.cpu().numpy(), B)


@pytest.mark.parametrize("Tin,Cin,Cout,k,s,B", [(203, 64, 96, 3, 2, 3), (101, 32, 32, 2, 2, 2), (49199 // 16, 512, 512, 3, 2, 1)])
def test_strided_conv_as_overlapping_gemm(env, Tin, Cin, Cout, k, s, B):
    """Conv1D(valid, stride) == GEMM over a window view with lda = s*Cin < k*Cin
    (feature_extractor.py:31-37), batched over samples."""
    lib, torch, dev = env
    x, w = rnd("x", (B, Tin, Cin)), rnd("w", (k, Cin, Cout), 0.1)
    bias = rnd("cb", (Cout,))
    ref = O.gelu(O.conv1d_valid(x.astype(np.float64), w.astype(np.float64), s, bias.astype(np.float64)))
    Tout = 1 + (Tin - k) // s
    tx, tw, tb = dev_t(torch, dev, x), dev_t(torch, dev, w), dev_t(torch, dev, bias)
    out = torch.empty((B, Tout, Cout), device=dev)
    N.check(lib.w2v2_op_gemm(N.ptr(tx), s * Cin, Tin * Cin, N.ptr(tw), Cout, N.ptr(out), Cout, Tout * Cout,
                             N.ptr(tb), None, Tout, Cout, k * Cin, B, 1, stream()))
    assert H.max_err(out.cpu().numpy(), ref) < 3e-5 * max(1.0, np.abs(ref).max())


def test_gemm_rejects_bad_arguments(env):
    lib, torch, dev = env
    t = torch.zeros((4, 4), device=dev)
    assert lib.w2v2_op_gemm(N.ptr(t), 4, 0, N.ptr(t), 4, N.ptr(t), 4, 0, None, None, 0, 4, 4, 1, 0, stream()) == -1
    assert b"gemm" in lib.w2v2_last_error()
    assert lib.w2v2_op_gemm(None, 4, 0, N.ptr(t), 4, N.ptr(t), 4, 0, None, None, 4, 4, 4, 1, 0, stream()) == -1


# ---------------------------------------------------------------- bf16-operand GEMM ----
@pytest.mark.parametrize("M,N_,K,act,use_bias,use_res", [
    (128, 128, 64, 0, False, False), (300, 130, 128, 1, True, False), (257, 32, 192, 0, True, True),
    (1000, 770, 100, 2, True, False), (65, 3, 7, 0, False, True), (2048, 768, 3072, 0, True, True),
    (515, 2304, 768, 0, True, False)])
def test_gemm_bf16_matches_rounded_operands(env, M, N_, K, act, use_bias, use_res):
    """W2V2_PRECISION_BF16's GEMM == exact products of nearest-even bf16 operands, wide accumulation, fp32
    epilogue.  Both the 16-byte fast path and the guarded path (odd K / N) are covered."""
    lib, torch, dev = env
    A, B = rnd("A16", (M, K)), rnd("B16", (K, N_), 0.2)
    bias = rnd("bias16", (N_,)) if use_bias else None
    res = rnd("res16", (M, N_)) if use_res else None
    ref = O.round_bf16(A).astype(np.float64) @ O.round_bf16(B).astype(np.float64)
    if use_bias:
        ref = ref + bias
    if act:
        ref = O.gelu(ref, approximate=(act == 2))
    if use_res:
        ref = ref + res
    tb = dev_t(torch, dev, bias) if use_bias else None
    tr = dev_t(torch, dev, res) if use_res else None
    out = torch.full((M, N_), float("nan"), device=dev)
    N.check(lib.w2v2_op_gemm_bf16(N.ptr(dev_t(torch, dev, A)), K, 0, N.ptr(dev_t(torch, dev, B)), N_, N.ptr(out), N_, 0,
                                  N.ptr(tb), N.ptr(tr), M, N_, K, 1, act, stream()))
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert H.max_err(got, ref) < 2e-5 * max(1.0, np.abs(ref).max())       # fp32 accumulation order only
    exact = A.astype(np.float64) @ B.astype(np.float64)
    assert H.max_err(O.round_bf16(A).astype(np.float64) @ O.round_bf16(B).astype(np.float64), exact) > 1e-4 or K < 16


def test_gemm_bf16_transpose_detecting(env):
    """A = I with an ASYMMETRIC bf16-exact B: a swapped k-pairing or C row/col map cannot pass."""
    lib, torch, dev = env
    n = 192
    A = np.eye(n, dtype=np.float32)
    B = ((np.arange(n)[:, None] % 16) * 16 + (np.arange(n)[None, :] % 13) + 1).astype(np.float32)   # < 256: exact in bf16
    B *= np.where(np.arange(n)[:, None] > np.arange(n)[None, :], 1.0, -1.0).astype(np.float32)
    out = torch.empty((n, n), device=dev)
    N.check(lib.w2v2_op_gemm_bf16(N.ptr(dev_t(torch, dev, A)), n, 0, N.ptr(dev_t(torch, dev, B)), n, N.ptr(out), n, 0,
                                  None, None, n, n, n, 1, 0, stream()))
    assert np.array_equal(out.cpu().numpy(), B)


def _bf16_bits(torch, t):
    """fp32 tensor -> its nearest-even bf16 copy (the shadow a producer kernel would write), as a device bf16 tensor."""
    return t.to(torch.bfloat16).contiguous()


@pytest.mark.parametrize("M,N_,K,act,use_bias,use_res,f32_out,b16_out", [
    (128, 256, 192, 0, False, False, True, True),        # smallest shape the software-pipelined kernel takes (3 K tiles)
    (300, 512, 448, 1, True, False, False, True),        # ragged rows, GELU, bf16-only output (the LDS-staged epilogue on full row tiles)
    (257, 768, 768, 0, True, True, True, False),         # fp32 output + residual (out-projection / FFN down form)
    (1000, 2304, 768, 2, True, False, True, True),       # tanh GELU, both outputs
    (640, 768, 3072, 0, True, True, True, True),         # long K
    (130, 256, 256, 1, True, False, False, True),
    (1300, 1536, 192, 0, True, True, True, True),        # 11 x 6 tiles: the grouped tile order (common.h::grouped_tile) with a short last group
    (2050, 2048, 256, 1, True, False, False, True),      # 17 x 8 tiles, ragged last row tile, bf16-only output
    (8192 + 128, 768, 768, 0, True, True, True, True)])  # 25-MB outputs: tile pointers whose low word has bit 31 set (a sign-extended
                                                         # scalar base of the residual prefetch faulted from M = 8192 on)
def test_gemm_bf16_shadow_kernels_agree_bit_for_bit(env, M, N_, K, act, use_bias, use_res, f32_out, b16_out):
    """The shadow-fed form of the bf16 GEMM (both operands already bf16, streamed HBM / L2 -> LDS by DMA) has two kernels:
    128 x 128 tiles (gemm_bf16.hip) and 128 x 256 software-pipelined 4-wave blocks (gemm_bf16_sw.hip).  Same products, same
    ascending k order, same lane -> k assignment: identical bits in the fp32 AND the bf16 output, and both equal to exact
    products of the bf16 operands up to fp32 accumulation order."""
    lib, torch, dev = env
    A, B = rnd("As16", (M, K)), rnd("Bs16", (K, N_), 0.2)
    bias = rnd("biass16", (N_,)) if use_bias else None
    res = rnd("ress16", (M, N_)) if use_res else None
    ref = O.round_bf16(A).astype(np.float64) @ O.round_bf16(B).astype(np.float64)
    if use_bias:
        ref = ref + bias
    if act:
        ref = O.gelu(ref, approximate=(act == 2))
    if use_res:
        ref = ref + res
    A16 = _bf16_bits(torch, dev_t(torch, dev, A))
    B16 = _bf16_bits(torch, dev_t(torch, dev, np.ascontiguousarray(B.T)))      # (N, K): the weight shadow's layout
    tb = dev_t(torch, dev, bias) if use_bias else None
    tr = dev_t(torch, dev, res) if use_res else None
    outs = {}
    for variant in (1, 2, 0):
        Cf = torch.full((M, N_), float("nan"), device=dev) if f32_out else None
        Ch = torch.zeros((M, N_), device=dev, dtype=torch.bfloat16) if b16_out else None
        N.check(lib.w2v2_op_gemm_bf16_shadows(N.ptr(A16), K, 0, N.ptr(B16), N.ptr(Cf), N.ptr(Ch), N_, 0, N.ptr(tb), N.ptr(tr), M, N_, K, 1, act,
                                              variant, stream()), "w2v2_op_gemm_bf16_shadows")
        torch.cuda.synchronize()
        outs[variant] = (Cf, Ch)
    for variant in (2, 0):
        if f32_out:
            assert torch.equal(outs[1][0], outs[variant][0]), f"fp32 output: variant {variant} differs from the 128 x 128 kernel"
        if b16_out:
            assert torch.equal(outs[1][1].view(torch.int16), outs[variant][1].view(torch.int16)), f"bf16 output: variant {variant} differs"
    scale = max(1.0, np.abs(ref).max())
    if f32_out:
        got = outs[2][0].cpu().numpy()
        assert np.isfinite(got).all() and H.max_err(got, ref) < 2e-5 * scale
    if b16_out:
        got = outs[2][1].float().cpu().numpy()
        assert H.max_err(got, O.round_bf16(ref.astype(np.float32))) <= 2.0 ** -7 * scale      # one bf16 step where the fp32 sums straddle a tie


def test_gemm_bf16_shadows_transpose_detecting_and_batched_conv(env):
    """(a) A = I against an asymmetric bf16-exact B on the software-pipelined kernel: a swapped k pairing, half-tile or C map
    cannot pass; (b) a strided Conv1D as an overlapping-row batched GEMM (lda < K, per-sample stride, ragged last row tile)."""
    lib, torch, dev = env
    n = 256
    A = np.eye(n, dtype=np.float32)
    Bm = ((np.arange(n)[:, None] % 16) * 16 + (np.arange(n)[None, :] % 13) + 1).astype(np.float32)   # < 256: exact in bf16
    Bm *= np.where(np.arange(n)[:, None] > np.arange(n)[None, :], 1.0, -1.0).astype(np.float32)
    A16 = _bf16_bits(torch, dev_t(torch, dev, A))
    B16 = _bf16_bits(torch, dev_t(torch, dev, np.ascontiguousarray(Bm.T)))
    out = torch.empty((n, n), device=dev)
    N.check(lib.w2v2_op_gemm_bf16_shadows(N.ptr(A16), n, 0, N.ptr(B16), N.ptr(out), None, n, 0, None, None, n, n, n, 1, 0, 2, stream()))
    assert np.array_equal(out.cpu().numpy(), Bm)
    Tin, Cin, Cout, k, st, Bn = 611, 128, 256, 3, 2, 3
    x, w = rnd("xs16", (Bn, Tin, Cin)), rnd("ws16", (k, Cin, Cout), 0.1)
    bias = rnd("cbs16", (Cout,))
    ref = O.gelu(O.conv1d_valid(O.round_bf16(x).astype(np.float64), O.round_bf16(w).astype(np.float64), st, bias.astype(np.float64)))
    Tout = 1 + (Tin - k) // st
    x16 = _bf16_bits(torch, dev_t(torch, dev, x))
    w16 = _bf16_bits(torch, dev_t(torch, dev, np.ascontiguousarray(w.reshape(k * Cin, Cout).T)))
    tb = dev_t(torch, dev, bias)
    got = {}
    for variant in (1, 2):
        o = torch.full((Bn, Tout, Cout), float("nan"), device=dev)
        N.check(lib.w2v2_op_gemm_bf16_shadows(N.ptr(x16), st * Cin, Tin * Cin, N.ptr(w16), N.ptr(o), None, Cout, Tout * Cout, N.ptr(tb), None,
                                              Tout, Cout, k * Cin, Bn, 1, variant, stream()))
        got[variant] = o
    assert torch.equal(got[1], got[2])
    assert H.max_err(got[2].cpu().numpy(), ref) < 3e-5 * max(1.0, np.abs(ref).max())
    # variant 2 on a shape the kernel does not take is an error, not a silent fallback
    assert lib.w2v2_op_gemm_bf16_shadows(N.ptr(x16), st * Cin, Tin * Cin, N.ptr(w16), N.ptr(got[1]), None, Cout, Tout * Cout, None, None,
                                         Tout, Cout, k * Cin, Bn, 0, 3, stream()) == -1


@pytest.mark.parametrize("Tin,Cin,Cout,k,s,B", [(203, 64, 96, 3, 2, 3), (101, 32, 32, 2, 2, 2), (49199 // 16, 512, 512, 3, 2, 1)])
def test_strided_conv_as_overlapping_gemm_bf16(env, Tin, Cin, Cout, k, s, B):
    lib, torch, dev = env
    x, w = rnd("x16", (B, Tin, Cin)), rnd("w16", (k, Cin, Cout), 0.1)
    bias = rnd("cb16", (Cout,))
    ref = O.gelu(O.conv1d_valid(O.round_bf16(x).astype(np.float64), O.round_bf16(w).astype(np.float64), s, bias.astype(np.float64)))
    Tout = 1 + (Tin - k) // s
    out = torch.empty((B, Tout, Cout), device=dev)
    N.check(lib.w2v2_op_gemm_bf16(N.ptr(dev_t(torch, dev, x)), s * Cin, Tin * Cin, N.ptr(dev_t(torch, dev, w)), Cout,
                                  N.ptr(out), Cout, Tout * Cout, N.ptr(dev_t(torch, dev, bias)), None, Tout, Cout, k * Cin,
                                  B, 1, stream()))
    assert H.max_err(out.cpu().numpy(), ref) < 3e-5 * max(1.0, np.abs(ref).max())



@pytest.mark.parametrize("M,N_,K,act,use_bias,use_res", [
    (128, 256, 32, 0, False, False), (300, 256, 128, 1, True, False), (1000, 768, 96, 0, True, True),
    (2048, 768, 3072, 0, True, True), (515, 2304, 768, 2, True, False), (2100, 512, 1536, 1, True, False)])   # (large enough that the fp32 kernel does not split K)
def test_gemm_split_is_fp32_grade(env, M, N_, K, act, use_bias, use_res):
    """W2V2_PRECISION_BF16X3's GEMM (csrc/gemm_split.hip): fp32 operands as exact three-term bf16 sums, six bf16 MFMA
    products per fp32 product, fp32 accumulation.  The claim is fp32-LEVEL accuracy, so the bar is the native fp32 MFMA
    kernel's: the same tolerance against fp64 as test_gemm_matches_numpy, and an error no larger than 1.5x what the
    fp32 kernel commits on the same operands (measured: 0.8-1.0x)."""
    lib, torch, dev = env
    A, B = rnd("As", (M, K)), rnd("Bs", (K, N_), 0.2)
    bias = rnd("biass", (N_,)) if use_bias else None
    res = rnd("ress", (M, N_)) if use_res else None
    ref = A.astype(np.float64) @ B.astype(np.float64)
    if use_bias:
        ref = ref + bias
    if act:
        ref = O.gelu(ref, approximate=(act == 2))
    if use_res:
        ref = ref + res
    tA, tB = dev_t(torch, dev, A), dev_t(torch, dev, B)
    tb = dev_t(torch, dev, bias) if use_bias else None
    tr = dev_t(torch, dev, res) if use_res else None
    out = torch.full((M, N_), float("nan"), device=dev)
    nat = torch.full((M, N_), float("nan"), device=dev)
    N.check(lib.w2v2_op_gemm_split(N.ptr(tA), K, 0, N.ptr(tB), N.ptr(out), N_, 0, N.ptr(tb), N.ptr(tr), M, N_, K, 1, act, stream()))
    N.check(lib.w2v2_op_gemm(N.ptr(tA), K, 0, N.ptr(tB), N_, N.ptr(nat), N_, 0, N.ptr(tb), N.ptr(tr), M, N_, K, 1, act, stream()))
    got, native = out.cpu().numpy(), nat.cpu().numpy()
    assert np.isfinite(got).all()
    scale = max(1.0, np.abs(ref).max())
    assert H.max_err(got, ref) < 2e-5 * scale
    rms = lambda x: float(np.sqrt(np.mean((x - ref) ** 2)))
    assert rms(got) <= 1.5 * rms(native) + 1e-9, (rms(got), rms(native))


def test_gemm_split_exact_on_three_term_operands(env):
    """The split itself is exact: operands whose fp32 significands use all 24 bits but whose products and sums stay
    exactly representable give the exact integer result (a two-term split would lose the low 8 bits of each operand)."""
    lib, torch, dev = env
    M, N_, K = 128, 256, 32
    idx = np.arange(M * K, dtype=np.int64).reshape(M, K)
    A = (((idx * 2654435761) % 4096) + 4096 * ((idx * 40503) % 4096)).astype(np.float32)      # 24-bit integers
    B = np.zeros((K, N_), np.float32)
    B[np.arange(N_) % K, np.arange(N_)] = 1.0                                                   # column n selects A[:, n % K]
    B[(np.arange(N_) + 1) % K, np.arange(N_)] = -1.0                                            # ... minus its neighbour
    ref = A.astype(np.float64) @ B.astype(np.float64)
    out = torch.empty((M, N_), device=dev)
    N.check(lib.w2v2_op_gemm_split(N.ptr(dev_t(torch, dev, A)), K, 0, N.ptr(dev_t(torch, dev, B)), N.ptr(out), N_, 0, None, None,
                                   M, N_, K, 1, 0, stream()))
    assert np.array_equal(out.cpu().numpy().astype(np.float64), ref)


def test_strided_conv_as_overlapping_gemm_split(env):
    lib, torch, dev = env
    Tin, Cin, Cout, k, s, B = 49199 // 16, 512, 512, 3, 2, 2
    x, w = rnd("xs", (B, Tin, Cin)), rnd("ws", (k, Cin, Cout), 0.1)
    bias = rnd("cbs", (Cout,))
    ref = O.gelu(O.conv1d_valid(x.astype(np.float64), w.astype(np.float64), s, bias.astype(np.float64)))
    Tout = 1 + (Tin - k) // s
    out = torch.empty((B, Tout, Cout), device=dev)
    N.check(lib.w2v2_op_gemm_split(N.ptr(dev_t(torch, dev, x)), s * Cin, Tin * Cin, N.ptr(dev_t(torch, dev, w)), N.ptr(out), Cout,
                                   Tout * Cout, N.ptr(dev_t(torch, dev, bias)), None, Tout, Cout, k * Cin, B, 1, stream()))
    assert H.max_err(out.cpu().numpy(), ref) < 2e-5 * max(1.0, np.abs(ref).max())


def test_gemm_split_rejects_unsupported_shapes(env):
    lib, torch, dev = env
    A, B = dev_t(torch, dev, rnd("Ar", (64, 32))), dev_t(torch, dev, rnd("Br", (32, 96)))
    out = torch.empty((64, 96), device=dev)
    assert lib.w2v2_op_gemm_split(N.ptr(A), 32, 0, N.ptr(B), N.ptr(out), 96, 0, None, None, 64, 96, 32, 1, 0, stream()) == -1   # N % 256


# ---- the plane-fed split GEMM (csrc/gemm_split_sw.hip): precision modes bf16x3 / f16x2 of the inference forward ----
PLANE_FMTS = {"bf16x3": (0, 3), "f16x2": (1, 2)}      # name -> (W2V2_PLANES_*, planes per element)
F16X2_ACT_SCALE = 16.0


def _planes_value(torch, p, n, fmt_name):
    """(planes * n) int16 device tensor -> the fp64 values the planes stand for"""
    if fmt_name == "f16x2":
        return (p[:n].view(torch.float16).double() + p[n:2 * n].view(torch.float16).double()) / F16X2_ACT_SCALE
    f = lambda q: (q.to(torch.int32) << 16).view(torch.float32).double()
    return f(p[:n]) + f(p[n:2 * n]) + f(p[2 * n:3 * n])


def _split_operands(lib, torch, dev, tA, tB, K, N_, fmt_name):
    fmt, npl = PLANE_FMTS[fmt_name]
    n = tA.numel()
    pA = torch.empty(npl * n, dtype=torch.int16, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    N.check(lib.w2v2_op_split_planes(N.ptr(tA), N.ptr(pA), n, n, fmt, N.ptr(flag), stream()))
    img = torch.empty(npl * K * N_, dtype=torch.int16, device=dev)
    ws = torch.zeros(2, dtype=torch.float32, device=dev)
    N.check(lib.w2v2_op_split_weight(N.ptr(tB), N.ptr(img), N.ptr(ws) if fmt else None, K, N_, fmt, stream()))
    _KEEP.extend([pA, img, ws, flag])
    return pA, img, (ws[1:] if fmt else None), flag


@pytest.mark.parametrize("fmt_name", ["bf16x3", "f16x2"])
def test_split_planes_represent_the_value(env, fmt_name):
    """What the producers write: bf16x3 planes sum back to the fp32 value EXACTLY; f16x2 planes (two fp16 terms of 16 x) to 2^-21
    relative (2^-28 absolute for tiny values: fp16 subnormals are kept), and a value beyond fp16's range sets the sticky flag."""
    lib, torch, dev = env
    x = np.concatenate([rnd("pl", (4096,), 3.0), rnd("pl2", (4096,), 1e-3), rnd("pl3", (1024,), 200.0), np.zeros(64, np.float32)]).astype(np.float32)
    t = dev_t(torch, dev, x)
    pA, _, _, flag = _split_operands(lib, torch, dev, t, dev_t(torch, dev, rnd("plw", (64, 256))), 64, 256, fmt_name)
    v = _planes_value(torch, pA, x.size, fmt_name).cpu().numpy()
    if fmt_name == "bf16x3":
        assert np.array_equal(v, x.astype(np.float64))
    else:
        assert (np.abs(v - x) <= np.abs(x) * 2.0 ** -21 + 2.0 ** -28).all()
        assert int(flag.item()) == 0
        big = dev_t(torch, dev, np.array([1.0, 5000.0, -2.0, 3.0], np.float32))
        p2 = torch.empty(8, dtype=torch.int16, device=dev)
        N.check(lib.w2v2_op_split_planes(N.ptr(big), N.ptr(p2), 4, 4, 1, N.ptr(flag), stream()))
        assert int(flag.item()) == 1 and bool(torch.isfinite(p2.view(torch.float16).float()).all())


@pytest.mark.parametrize("fmt_name", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("M,N_,K,act,use_bias,use_res", [
    (128, 256, 64, 0, False, False), (300, 256, 128, 1, True, False), (1000, 768, 192, 0, True, True), (1, 256, 64, 0, True, False),
    (2048, 768, 3072, 0, True, True), (515, 2304, 768, 2, True, False), (2100, 512, 1536, 1, True, False),
    (1300, 1536, 128, 0, True, True)])      # 11 x 6 tiles: the grouped tile order (common.h::grouped_tile) with a short last group
def test_gemm_split_planes_is_fp32_grade(env, fmt_name, M, N_, K, act, use_bias, use_res):
    """The plane-fed GEMM of precision modes bf16x3 (six bf16 products of exact three-term splits) and f16x2 (three fp16 products of
    two-term splits): same bar as test_gemm_split_is_fp32_grade -- the fp32 tolerance against fp64 and an rms error no larger than
    1.5x what the fp32 MFMA kernel commits on the same operands (measured 0.6-1.0x; f16x2 at K <= 128, where fp32 sums are still
    short, up to 1.1x) -- for fp32 output with residual, and for the plane output the next GEMM streams."""
    lib, torch, dev = env
    fmt, npl = PLANE_FMTS[fmt_name]
    A, B = rnd("Ap", (M, K)), rnd("Bp", (K, N_), 0.2)
    bias = rnd("biasp", (N_,)) if use_bias else None
    res = rnd("resp", (M, N_)) if use_res else None
    ref = A.astype(np.float64) @ B.astype(np.float64)
    if use_bias:
        ref = ref + bias
    if act:
        ref = O.gelu(ref, approximate=(act == 2))
    pre_res = ref
    if use_res:
        ref = ref + res
    tA, tB = dev_t(torch, dev, A), dev_t(torch, dev, B)
    tb = dev_t(torch, dev, bias) if use_bias else None
    tr = dev_t(torch, dev, res) if use_res else None
    pA, img, sc, flag = _split_operands(lib, torch, dev, tA, tB, K, N_, fmt_name)
    out = torch.full((M, N_), float("nan"), device=dev)
    nat = torch.full((M, N_), float("nan"), device=dev)
    N.check(lib.w2v2_op_gemm_split_planes(fmt, N.ptr(pA), M * K, K, 0, N.ptr(img), N.ptr(sc) if fmt else None, N.ptr(out), None, 0, N_, 0,
                                          N.ptr(tb), N.ptr(tr), M, N_, K, 1, act, N.ptr(flag), stream()))
    N.check(lib.w2v2_op_gemm(N.ptr(tA), K, 0, N.ptr(tB), N_, N.ptr(nat), N_, 0, N.ptr(tb), N.ptr(tr), M, N_, K, 1, act, stream()))
    got, native = out.cpu().numpy(), nat.cpu().numpy()
    assert np.isfinite(got).all()
    scale = max(1.0, np.abs(ref).max())
    assert H.max_err(got, ref) < 2e-5 * scale
    rms = lambda x, r: float(np.sqrt(np.mean((x - r) ** 2)))
    assert rms(got, ref) <= 1.5 * rms(native, ref) + 2e-8, (rms(got, ref), rms(native, ref))
    # the plane output (no residual in that form): bf16x3 planes ARE the fp32 result, f16x2 planes are it to 2^-21
    P = torch.full((npl * M * N_,), 0x7fff, dtype=torch.int16, device=dev)
    N.check(lib.w2v2_op_gemm_split_planes(fmt, N.ptr(pA), M * K, K, 0, N.ptr(img), N.ptr(sc) if fmt else None, None, N.ptr(P), M * N_, N_, 0,
                                          N.ptr(tb), None, M, N_, K, 1, act, N.ptr(flag), stream()))
    pv = _planes_value(torch, P, M * N_, fmt_name).cpu().numpy().reshape(M, N_)
    if not use_res:
        if fmt_name == "bf16x3":
            assert np.array_equal(pv, got.astype(np.float64))
        else:
            assert (np.abs(pv - got) <= np.abs(got) * 2.0 ** -20 + 2.0 ** -27).all()
    assert H.max_err(pv, pre_res) < 2e-5 * scale
    assert int(flag.item()) == 0


def test_gemm_split_planes_exact_on_three_term_operands(env):
    """bf16x3 through the plane-fed kernel keeps the exactness of the split (24-bit integer operands, exactly representable sums)."""
    lib, torch, dev = env
    M, N_, K = 128, 256, 64
    idx = np.arange(M * K, dtype=np.int64).reshape(M, K)
    A = (((idx * 2654435761) % 4096) + 4096 * ((idx * 40503) % 4096)).astype(np.float32)
    B = np.zeros((K, N_), np.float32)
    B[np.arange(N_) % K, np.arange(N_)] = 1.0
    B[(np.arange(N_) + 1) % K, np.arange(N_)] = -1.0
    ref = A.astype(np.float64) @ B.astype(np.float64)
    tA, tB = dev_t(torch, dev, A), dev_t(torch, dev, B)
    pA, img, _, flag = _split_operands(lib, torch, dev, tA, tB, K, N_, "bf16x3")
    out = torch.empty((M, N_), device=dev)
    N.check(lib.w2v2_op_gemm_split_planes(0, N.ptr(pA), M * K, K, 0, N.ptr(img), None, N.ptr(out), None, 0, N_, 0, None, None, M, N_, K, 1, 0,
                                          N.ptr(flag), stream()))
    assert np.array_equal(out.cpu().numpy().astype(np.float64), ref)


@pytest.mark.parametrize("fmt_name", ["bf16x3", "f16x2"])
def test_strided_conv_as_overlapping_gemm_split_planes(env, fmt_name):
    """Conv1D as a GEMM over overlapping plane rows, the batch in one launch, ragged last row tile, GELU, plane output."""
    lib, torch, dev = env
    fmt, npl = PLANE_FMTS[fmt_name]
    Tin, Cin, Cout, k, s, B = 49199 // 16, 512, 512, 3, 2, 2
    x, w = rnd("xp", (B, Tin, Cin)), rnd("wp", (k, Cin, Cout), 0.1)
    bias = rnd("cbp", (Cout,))
    ref = O.gelu(O.conv1d_valid(x.astype(np.float64), w.astype(np.float64), s, bias.astype(np.float64)))
    Tout = 1 + (Tin - k) // s
    tx, tw = dev_t(torch, dev, x), dev_t(torch, dev, w)
    px, img, sc, flag = _split_operands(lib, torch, dev, tx, tw, k * Cin, Cout, fmt_name)
    out = torch.empty((B, Tout, Cout), device=dev)
    N.check(lib.w2v2_op_gemm_split_planes(fmt, N.ptr(px), x.size, s * Cin, Tin * Cin, N.ptr(img), N.ptr(sc) if fmt else None, N.ptr(out), None, 0, Cout,
                                          Tout * Cout, N.ptr(dev_t(torch, dev, bias)), None, Tout, Cout, k * Cin, B, 1, N.ptr(flag), stream()))
    assert H.max_err(out.cpu().numpy(), ref) < 2e-5 * max(1.0, np.abs(ref).max())
    P = torch.empty(npl * B * Tout * Cout, dtype=torch.int16, device=dev)
    N.check(lib.w2v2_op_gemm_split_planes(fmt, N.ptr(px), x.size, s * Cin, Tin * Cin, N.ptr(img), N.ptr(sc) if fmt else None, None, N.ptr(P), B * Tout * Cout,
                                          Cout, Tout * Cout, N.ptr(dev_t(torch, dev, bias)), None, Tout, Cout, k * Cin, B, 1, N.ptr(flag), stream()))
    pv = _planes_value(torch, P, B * Tout * Cout, fmt_name).cpu().numpy().reshape(B, Tout, Cout)
    assert H.max_err(pv, ref) < 2e-5 * max(1.0, np.abs(ref).max())


def test_gemm_split_planes_rejects_unsupported_shapes(env):
    lib, torch, dev = env
    p = torch.zeros(3 * 64 * 96, dtype=torch.int16, device=dev)
    out = torch.empty((64, 96), device=dev)
    assert lib.w2v2_op_gemm_split_planes(0, N.ptr(p), 64 * 64, 64, 0, N.ptr(p), None, N.ptr(out), None, 0, 96, 0, None, None, 64, 96, 64, 1, 0, None, stream()) == -1   # N % 256
    out2 = torch.empty((64, 256), device=dev)
    assert lib.w2v2_op_gemm_split_planes(0, N.ptr(p), 64 * 32, 32, 0, N.ptr(p), None, N.ptr(out2), None, 0, 256, 0, None, None, 64, 256, 32, 1, 0, None, stream()) == -1   # K % 64
    assert lib.w2v2_op_gemm_split_planes(1, N.ptr(p), 64 * 64, 64, 0, N.ptr(p), None, N.ptr(out2), None, 0, 256, 0, None, None, 64, 256, 64, 1, 0, None, stream()) == -1   # f16x2 without its scale
    assert lib.w2v2_op_split_planes(N.ptr(out), N.ptr(p), 64 * 96, 64 * 96, 7, None, stream()) == -1                                                                       # unknown format


def test_select_form_gelu_is_the_library_function(env):
    """csrc/common.h: erf_select / tanh_select (the device library's erff / tanhf with both arms evaluated and the result selected:
    no control flow around the split GEMM's 128 accumulators) return the library's bits for EVERY one of the 2^32 float patterns."""
    lib, torch, dev = env
    mm = torch.full((2,), -1, dtype=torch.int64, device=dev)
    N.check(lib.w2v2_op_check_select_forms(N.ptr(mm), stream()))
    assert mm.tolist() == [0, 0]


@pytest.mark.parametrize("rows,Kin,Nout,per,S", [(512, 128, 256, 128, 4), (1499 * 2, 256, 128, 1024, 3), (23984, 128, 384, 12032, 2),
                                                 (65, 128, 128, 64, 2), (37, 128, 128, 64, 1)])
def test_weight_grad_bf16_ragged_rows(env, rows, Kin, Nout, per, S):
    """dW = X^T dY from the bf16 copies of both operands (LDS-DMA + transposing LDS reads), S slabs of `per` rows: the row count
    need not fill the last slab or be a multiple of the 64-row K tile (B T = 23984 at 16 x 480000 samples) -- rows past it count
    as zero, whatever the memory behind the operands holds (here: NaN patterns)."""
    lib, torch, dev = env
    X, dY = rnd("wgX", (rows, Kin)), rnd("wgY", (rows, Nout), 0.3)
    Xr, Yr = O.round_bf16(X), O.round_bf16(dY)
    def dev16(a, pad_rows):      # bf16 bit patterns, followed by rows of NaN
        bits = (a.view(np.uint32) >> 16).astype(np.uint16)
        full = np.concatenate([bits, np.full((pad_rows, a.shape[1]), 0x7FC0, np.uint16)])
        return torch.from_numpy(full.view(np.int16)).to(dev)
    pad = per * S - rows
    assert 0 <= pad < per
    x16, y16 = dev16(Xr, pad), dev16(Yr, pad)
    out = torch.full((S, Kin, Nout), float("nan"), device=dev)
    N.check(lib.w2v2_op_weight_grad_bf16(N.ptr(x16), N.ptr(y16), N.ptr(out), rows, Kin, Nout, per, S, 0, stream()))
    got = out.cpu().numpy()
    for z in range(S):
        ref = Xr[z * per:(z + 1) * per].astype(np.float64).T @ Yr[z * per:(z + 1) * per].astype(np.float64)
        assert H.max_err(got[z], ref) < 2e-5 * max(1.0, np.abs(ref).max()), z
    # an empty slab is an error, not silent zeros
    assert lib.w2v2_op_weight_grad_bf16(N.ptr(x16), N.ptr(y16), N.ptr(out), max(1, per * (S - 1)), Kin, Nout, per, S, 0, stream()) == -1 or S == 1


@pytest.mark.parametrize("rows,Kin,Nout,S", [(192, 128, 256, 1), (1024, 768, 768, 4), (1536, 256, 2304, 3), (2048, 3072, 768, 2)])
def test_weight_grad_bf16_kernels_agree_bit_for_bit(env, rows, Kin, Nout, S):
    """dW = X^T dY from the bf16 copies of both operands has two kernels: 128 x 128 tiles (gemm_bf16_tr_kernel) and the 128 x 256
    software-pipelined kernel in its transposed form (k-major items, ds_read_b64_tr_b16 fragments).  Identical bits per slab, both
    equal to the exact products of the bf16 operands up to fp32 accumulation order; an asymmetric pattern catches a swapped
    operand, k half or column group."""
    lib, torch, dev = env
    X, dY = rnd("wg2X", (rows, Kin)), rnd("wg2Y", (rows, Nout), 0.3)
    X[:, 5] += 3.0; dY[:, Nout - 7] -= 2.0                        # mark one row of dW^T and one column
    Xr, Yr = O.round_bf16(X), O.round_bf16(dY)
    x16 = torch.from_numpy((Xr.view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).to(dev)
    y16 = torch.from_numpy((Yr.view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).to(dev)
    per = rows // S
    outs = {}
    for variant in (1, 2, 0):
        out = torch.full((S, Kin, Nout), float("nan"), device=dev)
        N.check(lib.w2v2_op_weight_grad_bf16(N.ptr(x16), N.ptr(y16), N.ptr(out), rows, Kin, Nout, per, S, variant, stream()), "w2v2_op_weight_grad_bf16")
        torch.cuda.synchronize()
        outs[variant] = out
    assert torch.equal(outs[1], outs[2]) and torch.equal(outs[1], outs[0])
    got = outs[2].cpu().numpy()
    for z in range(S):
        ref = Xr[z * per:(z + 1) * per].astype(np.float64).T @ Yr[z * per:(z + 1) * per].astype(np.float64)
        assert H.max_err(got[z], ref) < 2e-5 * max(1.0, np.abs(ref).max()), z
    # variant 2 cannot take ragged rows: an error, not a silent fallback
    assert lib.w2v2_op_weight_grad_bf16(N.ptr(x16), N.ptr(y16), N.ptr(outs[0]), rows - 8, Kin, Nout, per, S, 2, stream()) == -1


@pytest.mark.parametrize("q,e,Kin,Nout,S", [(3, 1, 128, 256, 2), (5, 4, 768, 768, 7), (4, 2, 256, 2304, 3)])
def test_weight_grad_bf16_uneven_slabs(env, q, e, Kin, Nout, S):
    """The training step cuts the B T rows of dW = X^T dY into as many slabs as fill the chip's block slots, whatever the row count's
    divisors: the first e slabs run one 64-row K tile more than the others (GemmShadows::kextra, 128 x 256 transposed kernel).  Every
    slab must equal the exact product over ITS rows (a slab that started at the wrong row cannot), and the slabs must add up to the
    whole product."""
    lib, torch, dev = env
    rows = 64 * (S * q + e)
    X, dY = rnd("wg3X", (rows, Kin)), rnd("wg3Y", (rows, Nout), 0.3)
    X[:, 3] += 2.0; dY[:, Nout - 5] -= 1.5
    X *= (1.0 + np.arange(rows, dtype=np.float32) / rows)[:, None]            # rows differ in scale: a shifted slab shows
    Xr, Yr = O.round_bf16(X), O.round_bf16(dY)
    x16 = torch.from_numpy((Xr.view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).to(dev)
    y16 = torch.from_numpy((Yr.view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).to(dev)
    out = torch.full((S, Kin, Nout), float("nan"), device=dev)
    N.check(lib.w2v2_op_weight_grad_bf16(N.ptr(x16), N.ptr(y16), N.ptr(out), rows, Kin, Nout, 64 * q, S, 2, stream()), "w2v2_op_weight_grad_bf16")
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    start = 0
    for z in range(S):
        n = 64 * (q + (1 if z < e else 0))
        ref = Xr[start:start + n].astype(np.float64).T @ Yr[start:start + n].astype(np.float64)
        assert np.isfinite(got[z]).all() and H.max_err(got[z], ref) < 2e-5 * max(1.0, np.abs(ref).max()), z
        start += n
    assert start == rows
    # e >= S is not a split this form describes
    assert lib.w2v2_op_weight_grad_bf16(N.ptr(x16), N.ptr(y16), N.ptr(out), 64 * (S * q + S), Kin, Nout, 64 * q, S, 2, stream()) == -1


@pytest.mark.parametrize("rows,Kin,Nout,S", [(3 * 64 + 1, 128, 256, 1), (23 * 64 + 48, 768, 768, 4), (14 * 64 + 63, 256, 2304, 3), (374 * 64 + 48, 1024, 1024, 16)])
def test_weight_grad_bf16_ragged_last_k_tile(env, rows, Kin, Nout, S):
    """B T need not be a multiple of the 64-row K tile (T = 1499 at 480000 samples: B T = 23984).  With an all-zero row kept behind
    dy16 the 128 x 256 transposed kernel reads the last tile's missing rows from there (and from x16's last row, which must not leak
    into the result): variant 3.  Per slab it must equal the exact product over its own rows, and -- on slab boundaries both kernels
    can express -- carry the bits of the 128 x 128 kernel, which reads missing rows as zero by address select."""
    lib, torch, dev = env
    X, dY = rnd("wg4X", (rows, Kin)), rnd("wg4Y", (rows + 1, Nout), 0.3)
    X[-1] = 1000.0                                   # the row the kernel re-reads for the missing ones: a leak would be loud
    dY[-1] = 0.0                                     # the promised zero row
    Xr, Yr = O.round_bf16(X), O.round_bf16(dY)
    x16 = torch.from_numpy((Xr.view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).to(dev)
    y16 = torch.from_numpy((Yr.view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).to(dev)
    units = (rows + 63) // 64
    q, e = units // S, units % S
    out3 = torch.full((S, Kin, Nout), float("nan"), device=dev)
    N.check(lib.w2v2_op_weight_grad_bf16(N.ptr(x16), N.ptr(y16), N.ptr(out3), rows, Kin, Nout, 64 * q, S, 3, stream()), "w2v2_op_weight_grad_bf16")
    torch.cuda.synchronize()
    got = out3.cpu().numpy()
    start = 0
    for z in range(S):
        n = min(64 * (q + (1 if z < e else 0)), rows - start)
        ref = Xr[start:start + n].astype(np.float64).T @ Yr[start:start + n].astype(np.float64)
        assert np.isfinite(got[z]).all() and H.max_err(got[z], ref) < 2e-5 * max(1.0, np.abs(ref).max()), z
        start += n
    assert start == rows
    if e == 0:          # even slabs: the 128 x 128 kernel (rows past the end read as zero) must give the same bits
        out1 = torch.full((S, Kin, Nout), float("nan"), device=dev)
        N.check(lib.w2v2_op_weight_grad_bf16(N.ptr(x16), N.ptr(y16), N.ptr(out1), rows, Kin, Nout, 64 * q, S, 1, stream()), "w2v2_op_weight_grad_bf16")
        torch.cuda.synchronize()
        assert torch.equal(out1, out3)
    # without the promise (variant 2) ragged rows are refused
    assert lib.w2v2_op_weight_grad_bf16(N.ptr(x16), N.ptr(y16), N.ptr(out3), rows, Kin, Nout, 64 * q, S, 2, stream()) == -1


@pytest.mark.parametrize("rows,Kin,Nout,S", [(256, 128, 132, 1), (512, 768, 64, 4), (1024, 132, 256, 8)])
def test_gemm_bf16_transposed_a_split_k(env, rows, Kin, Nout, S):
    """dW = X^T dY as the training step runs it in precision mode 1: X (rows, Kin) is passed as the TRANSPOSED A of the GEMM
    (no transposed copy), split over S batches along the rows, one (Kin, Nout) slab per batch."""
    lib, torch, dev = env
    X, dY = rnd("atX", (rows, Kin)), rnd("atY", (rows, Nout), 0.3)
    Kp = rows // S
    out = torch.full((S, Kin, Nout), float("nan"), device=dev)
    N.check(lib.w2v2_op_gemm_bf16_at(N.ptr(dev_t(torch, dev, X)), Kin, Kp * Kin, N.ptr(dev_t(torch, dev, dY)), Nout, Kp * Nout,
                                     N.ptr(out), Nout, Kin * Nout, Kin, Nout, Kp, S, stream()))
    got = out.cpu().numpy()
    Xr, Yr = O.round_bf16(X).astype(np.float64), O.round_bf16(dY).astype(np.float64)
    for z in range(S):
        ref = Xr[z * Kp:(z + 1) * Kp].T @ Yr[z * Kp:(z + 1) * Kp]
        assert H.max_err(got[z], ref) < 2e-5 * max(1.0, np.abs(ref).max()), z
    assert H.max_err(got.sum(0), Xr.T @ Yr) < 1e-4 * max(1.0, np.abs(Xr.T @ Yr).max())


# ---------------------------------------------------------------- LayerNorm -----
@pytest.mark.parametrize("rows,Cn,act", [(37, 512, 0), (5, 768, 1), (130, 1024, 0), (9, 64, 0), (3, 50, 0), (4, 1500, 1)])
def test_layer_norm(env, rows, Cn, act):
    lib, torch, dev = env
    x = rnd("lnx", (rows, Cn), 3.0) + 1.5
    g, b = 1 + rnd("lng", (Cn,), 0.2), rnd("lnb", (Cn,), 0.2)
    ref = O.layer_norm(x.astype(np.float64), g.astype(np.float64), b.astype(np.float64), 1e-5)
    if act:
        ref = O.gelu(ref)
    out = torch.empty((rows, Cn), device=dev)
    N.check(lib.w2v2_op_layer_norm(N.ptr(dev_t(torch, dev, x)), N.ptr(out), N.ptr(dev_t(torch, dev, g)),
                                   N.ptr(dev_t(torch, dev, b)), rows, Cn, 1e-5, act, stream()))
    assert H.max_err(out.cpu().numpy(), ref) < 1e-5


# ---------------------------------------------------------------- conv0 ---------
@pytest.mark.parametrize("B,L,Cn,K,S,bias", [(2, 4000, 32, 10, 5, False), (1, 46797, 512, 10, 5, False),
                                             (2, 1003, 48, 7, 3, True), (1, 10, 512, 10, 5, False)])
def test_conv0_groupnorm_gelu(env, B, L, Cn, K, S, bias):
    """conv0 -> per-(sample, channel) norm over time -> GELU, fused (feature_extractor.py:31-47;
    tensorflow_addons.py:207-231)."""
    lib, torch, dev = env
    x = V.hash_normal("c0x", B * L, 5).reshape(B, L)
    if L > 20000:
        x[:, L // 3:] = 0.0                       # padded tail, as in the 246000 convention
    w = rnd("c0w", (K, 1, Cn), 0.6)
    bs = rnd("c0b", (Cn,)) if bias else None
    g, b = 1 + rnd("c0g", (Cn,), 0.1), rnd("c0be", (Cn,), 0.1)
    y = O.conv1d_valid(x.astype(np.float64)[:, :, None], w.astype(np.float64), S, None if bs is None else bs.astype(np.float64))
    ref = O.gelu(O.group_norm_time(y, g.astype(np.float64), b.astype(np.float64), 1e-5))
    T0 = 1 + (L - K) // S
    out = torch.full((B, T0, Cn), float("nan"), device=dev)
    ws = torch.empty((int(lib.w2v2_conv0_ws_floats(B, L, K, S, Cn)),), device=dev)
    N.check(lib.w2v2_op_conv0(N.ptr(dev_t(torch, dev, x)), N.ptr(dev_t(torch, dev, w)),
                              N.ptr(dev_t(torch, dev, bs)) if bias else None, N.ptr(dev_t(torch, dev, g)),
                              N.ptr(dev_t(torch, dev, b)), N.ptr(out), N.ptr(ws), B, L, K, S, Cn, 1e-5, 0, 1, stream()))
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    # T0 == 1 is degenerate: var = 0, so rsqrt(eps) = 316 multiplies the fp32 rounding of the
    # scale/shift form  y*inv + (beta - mean*inv)  that tf.nn.batch_normalization (and this kernel) use
    tol = 1e-4 if T0 == 1 else 2e-5 * max(1.0, np.abs(ref).max())
    assert H.max_err(got, ref) < tol


@pytest.mark.parametrize("B,L,Cn,S,bias", [(2, 3000, 512, 5, True), (1, 4005, 64, 5, False), (3, 400, 32, 5, True), (2, 1000, 48, 3, True)])
def test_conv0_layernorm_gelu(env, B, L, Cn, S, bias):
    """conv0 -> LayerNormalization over channels -> GELU in one pass (the robust / xlsr extractor, feature_extractor.py:40-50):
    frame moments from the 10 input samples and the kernel's 10 x 10 moment matrix.  stride 3 takes the two-kernel fallback.
    A loud frame next to silence (padding) checks that the variance form has nothing to cancel."""
    lib, torch, dev = env
    K = 10
    x = V.hash_normal("l0x", B * L, 5).reshape(B, L)
    x[:, L // 2:L // 2 + 200] *= 50.0
    x[:, (3 * L) // 4:] = 0.0
    w = rnd("l0w", (K, 1, Cn), 0.6)
    bs = rnd("l0b", (Cn,)) if bias else None
    g, b = 1 + rnd("l0g", (Cn,), 0.1), rnd("l0be", (Cn,), 0.1)
    y = O.conv1d_valid(x.astype(np.float64)[:, :, None], w.astype(np.float64), S, None if bs is None else bs.astype(np.float64))
    ref = O.gelu(O.layer_norm(y, g.astype(np.float64), b.astype(np.float64), 1e-5))
    T0 = 1 + (L - K) // S
    out = torch.full((B, T0, Cn), float("nan"), device=dev)
    ws = torch.empty((int(lib.w2v2_conv0_ws_floats(B, L, K, S, Cn)),), device=dev)
    N.check(lib.w2v2_op_conv0(N.ptr(dev_t(torch, dev, x)), N.ptr(dev_t(torch, dev, w)),
                              N.ptr(dev_t(torch, dev, bs)) if bias else None, N.ptr(dev_t(torch, dev, g)),
                              N.ptr(dev_t(torch, dev, b)), N.ptr(out), N.ptr(ws), B, L, K, S, Cn, 1e-5, 2, 1, stream()))
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    # frames of pure padding without a bias have zero variance: rsqrt(eps) = 316 multiplies the fp32 rounding of the taps
    assert H.max_err(got, ref) < 2e-5 * max(1.0, np.abs(ref).max())


def test_conv0_plain_mode(env):
    lib, torch, dev = env
    B, L, Cn = 2, 3000, 64
    x, w, bs = rnd("p0x", (B, L)), rnd("p0w", (10, 1, Cn), 0.5), rnd("p0b", (Cn,))
    ref = O.conv1d_valid(x[:, :, None].astype(np.float64), w.astype(np.float64), 5, bs.astype(np.float64))
    out = torch.empty((B, ref.shape[1], Cn), device=dev)
    N.check(lib.w2v2_op_conv0(N.ptr(dev_t(torch, dev, x)), N.ptr(dev_t(torch, dev, w)), N.ptr(dev_t(torch, dev, bs)),
                              None, None, N.ptr(out), None, B, L, 10, 5, Cn, 1e-5, 1, 0, stream()))
    assert H.max_err(out.cpu().numpy(), ref) < 1e-5


# ---------------------------------------------------------------- pos conv ------
@pytest.mark.parametrize("B,T,Hh,K,G,flen", [(2, 12, 64, 16, 4, None), (2, 145, 768, 128, 16, None),
                                              (2, 200, 1024, 128, 16, [200, 130]), (1, 300, 128, 15, 4, [257])])
def test_pos_conv_weight_norm(env, B, T, Hh, K, G, flen):
    """weight-norm (per tap) + pad K/2 + grouped conv + drop-last-if-even + GELU + residual
    (tensorflow_addons.py:16-21,50-53; encoder.py:177-181,253,265)."""
    lib, torch, dev = env
    cg = Hh // G
    x = rnd("pcx", (B, T, Hh))
    wv, wg, bias = rnd("pcv", (K, cg, Hh), 0.3), 0.5 + rnd("pcg", (K, 1, 1), 0.3) ** 2, rnd("pcb", (Hh,), 0.1)
    xz = x.astype(np.float64).copy()
    if flen is not None:
        for b in range(B):
            xz[b, flen[b]:] = 0.0
    kern = O.weight_norm_kernel(wv.astype(np.float64), wg.astype(np.float64))
    y = O.grouped_conv1d_same(xz, kern, bias.astype(np.float64), G, K // 2)
    if K % 2 == 0:
        y = y[:, :-1]
    ref = xz + O.gelu(y)
    twg = torch.empty((G, K, cg, cg), device=dev)
    N.check(lib.w2v2_op_weight_norm_regroup(N.ptr(dev_t(torch, dev, wv)), N.ptr(dev_t(torch, dev, wg)), N.ptr(twg),
                                            K, cg, Hh, G, stream()))
    ref_w = np.stack([kern[:, :, g * cg:(g + 1) * cg] for g in range(G)])
    assert H.max_err(twg.cpu().numpy(), ref_w) < 1e-6
    tf = dev_t(torch, dev, np.asarray(flen, dtype=np.int32)) if flen is not None else None
    out = torch.full((B, T, Hh), float("nan"), device=dev)
    N.check(lib.w2v2_op_pos_conv(N.ptr(dev_t(torch, dev, x)), N.ptr(twg), N.ptr(dev_t(torch, dev, bias)), N.ptr(tf),
                                 N.ptr(out), B, T, Hh, K, G, 1, stream()))
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert H.max_err(got, ref) < 3e-5


# ---------------------------------------------------------------- attention -----
@pytest.mark.parametrize("B,T,Hh,heads,flen", [(2, 12, 64, 2, None), (1, 145, 768, 12, None), (2, 768, 128, 2, None),
                                                (2, 200, 128, 2, [200, 61]), (1, 97, 64, 2, [0])])
def test_attention(env, B, T, Hh, heads, flen):
    """softmax((q d^-0.5) k^T + key mask) v per head (encoder.py:22-47, 256-263)."""
    lib, torch, dev = env
    d = Hh // heads
    qkv = rnd("qkv", (B, T, 3 * Hh), 2.0)
    q, k, v = [qkv[:, :, i * Hh:(i + 1) * Hh].astype(np.float64).reshape(B, T, heads, d).transpose(0, 2, 1, 3) for i in range(3)]
    s = (q * d ** -0.5) @ k.transpose(0, 1, 3, 2)
    if flen is not None:
        keep = np.arange(T)[None, :] < np.asarray(flen)[:, None]
        # the reference adds the -10000 mask in fp32 (encoder.py:256-257): ulp(1e4) = 1e-3 quantises the
        # masked scores; emulate that rounding so fully-masked rows compare like with like
        s = (s.astype(np.float32) + ((1.0 - keep) * -10000.0)[:, None, None, :].astype(np.float32)).astype(np.float64)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    ref = (p @ v).transpose(0, 2, 1, 3).reshape(B, T, Hh)
    tf = dev_t(torch, dev, np.asarray(flen, dtype=np.int32)) if flen is not None else None
    out = torch.full((B, T, Hh), float("nan"), device=dev)
    N.check(lib.w2v2_op_attention(N.ptr(dev_t(torch, dev, qkv)), N.ptr(tf), N.ptr(out), B, T, Hh, heads, stream()))
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert H.max_err(got, ref) < 2e-5


@pytest.fixture
def bf16_ops(env):
    """Per-kernel calls in W2V2_PRECISION_BF16 for the duration of one test."""
    lib = env[0]
    N.check(lib.w2v2_op_set_precision(1))
    yield
    N.check(lib.w2v2_op_set_precision(0))


@pytest.mark.parametrize("B,T,Hh,heads,flen", [(1, 145, 768, 12, None), (2, 768, 128, 2, None), (2, 200, 128, 2, [200, 61]),
                                                (1, 97, 64, 1, [0]), (2, 64, 64, 1, None), (1, 33, 128, 2, None)])
def test_attention_bf16(env, bf16_ops, B, T, Hh, heads, flen):
    """bf16 matrix-pipe attention (head size 64): q d^-0.5, k, v and the probabilities enter the two contractions
    rounded to bf16, everything else fp32.  Checked against the fp64 formula on bf16-rounded q, k, v; what is
    left is the rounding of P (2^-9 relative per term, averaging out over the keys) -- a layout or key-order
    mistake would show up at O(1)."""
    lib, torch, dev = env
    d = Hh // heads
    assert d == 64
    qkv = rnd("qkv16", (B, T, 3 * Hh), 2.0)
    q, k, v = [qkv[:, :, i * Hh:(i + 1) * Hh].reshape(B, T, heads, d).transpose(0, 2, 1, 3) for i in range(3)]
    q = O.round_bf16(q * np.float32(d ** -0.5)).astype(np.float64)
    k, v = O.round_bf16(k).astype(np.float64), O.round_bf16(v).astype(np.float64)
    s = q @ k.transpose(0, 1, 3, 2)
    if flen is not None:
        keep = np.arange(T)[None, :] < np.asarray(flen)[:, None]
        s = (s.astype(np.float32) + ((1.0 - keep) * -10000.0)[:, None, None, :].astype(np.float32)).astype(np.float64)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    ref = (p @ v).transpose(0, 2, 1, 3).reshape(B, T, Hh)
    tf = dev_t(torch, dev, np.asarray(flen, dtype=np.int32)) if flen is not None else None
    out = torch.full((B, T, Hh), float("nan"), device=dev)
    N.check(lib.w2v2_op_attention(N.ptr(dev_t(torch, dev, qkv)), N.ptr(tf), N.ptr(out), B, T, Hh, heads, stream()))
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    err = np.abs(got - ref)
    print(f"bf16 attention: max err {err.max():.3e}, mean err {err.mean():.3e}, max|ref| {np.abs(ref).max():.3f}")
    assert err.max() < 2e-2 and err.mean() < 2e-3


@pytest.fixture(params=[2, 3], ids=["bf16x3", "f16x2"])
def bf16x3_ops(env, request):
    """Per-kernel calls in W2V2_PRECISION_BF16X3 / W2V2_PRECISION_F16X2 for the duration of one test."""
    lib = env[0]
    N.check(lib.w2v2_op_set_precision(request.param))
    yield
    N.check(lib.w2v2_op_set_precision(0))


@pytest.mark.parametrize("B,T,Hh,heads,flen", [(1, 145, 768, 12, None), (2, 768, 128, 2, None), (2, 200, 128, 2, [200, 61]),
                                                (1, 97, 64, 1, [0]), (2, 64, 64, 1, None), (1, 33, 128, 2, None), (1, 300, 64, 1, [257])])
def test_attention_split_is_fp32_grade(env, bf16x3_ops, B, T, Hh, heads, flen):
    """The attention of precision modes bf16x3 / f16x2 (csrc/attention_split.hip, head size 64): q d^-0.5, k, v and the probabilities
    as exact three-term bf16 sums with six MFMA products per fp32 product, or as two-term fp16 sums with three.  Same fp64 formula
    and the SAME 2e-5 bound as the fp32 kernel's test_attention, plus: no worse than 1.5x the fp32 kernel's error on the same input."""
    lib, torch, dev = env
    d = Hh // heads
    assert d == 64
    qkv = rnd("qkv", (B, T, 3 * Hh), 2.0)
    q, k, v = [qkv[:, :, i * Hh:(i + 1) * Hh].astype(np.float64).reshape(B, T, heads, d).transpose(0, 2, 1, 3) for i in range(3)]
    s = (q * d ** -0.5) @ k.transpose(0, 1, 3, 2)
    if flen is not None:
        keep = np.arange(T)[None, :] < np.asarray(flen)[:, None]
        s = (s.astype(np.float32) + ((1.0 - keep) * -10000.0)[:, None, None, :].astype(np.float32)).astype(np.float64)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    ref = (p @ v).transpose(0, 2, 1, 3).reshape(B, T, Hh)
    tf = dev_t(torch, dev, np.asarray(flen, dtype=np.int32)) if flen is not None else None
    tq = dev_t(torch, dev, qkv)
    out = torch.full((B, T, Hh), float("nan"), device=dev)
    N.check(lib.w2v2_op_attention(N.ptr(tq), N.ptr(tf), N.ptr(out), B, T, Hh, heads, stream()))
    got = out.cpu().numpy()
    N.check(lib.w2v2_op_set_precision(0))
    nat = torch.full((B, T, Hh), float("nan"), device=dev)
    N.check(lib.w2v2_op_attention(N.ptr(tq), N.ptr(tf), N.ptr(nat), B, T, Hh, heads, stream()))
    native = nat.cpu().numpy()
    assert np.isfinite(got).all()
    e3, e32 = H.max_err(got, ref), H.max_err(native, ref)
    print(f"split attention: max err {e3:.3e} (fp32 kernel {e32:.3e})")
    assert not np.array_equal(got, native)          # a different kernel did run
    assert e3 < 2e-5
    assert e3 < 1.5 * e32 + 1e-6


def test_attention_softmax_spike(env):
    """One key dominating by ~60 nats mid-sequence: forces the online-softmax rescale."""
    lib, torch, dev = env
    B, T, Hh, heads = 1, 256, 64, 1
    qkv = rnd("spk", (B, T, 3 * Hh), 1.0)
    qkv[0, 5, :Hh] = 6.0            # query 5
    qkv[0, 150, Hh:2 * Hh] = 10.0   # key 150 aligned with it
    q, k, v = [qkv[0, :, i * Hh:(i + 1) * Hh].astype(np.float64) for i in range(3)]
    s = (q * Hh ** -0.5) @ k.T
    p = np.exp(s - s.max(-1, keepdims=True))
    ref = (p / p.sum(-1, keepdims=True)) @ v
    out = torch.empty((B, T, Hh), device=dev)
    N.check(lib.w2v2_op_attention(N.ptr(dev_t(torch, dev, qkv)), None, N.ptr(out), B, T, Hh, heads, stream()))
    assert H.max_err(out.cpu().numpy()[0], ref) < 2e-5


# ---------------------------------------------------------------- lengths / CTC --
def test_frame_lengths(env):
    lib, torch, dev = env
    cfg = H.case_config("base_sample_padded")
    L = 246000
    m = np.ones((4, L), np.int32)
    m[1, 46797:] = 0
    m[2, 400:] = 0
    m[3, :] = 0
    ks = (C.c_int32 * 7)(*cfg.kernal_sizes)
    ss = (C.c_int32 * 7)(*cfg.strides)
    out = torch.empty((4,), device=dev, dtype=torch.int32)
    N.check(lib.w2v2_op_frame_lengths(N.ptr(dev_t(torch, dev, m)), N.ptr(out), 4, L, ks, ss, 7, stream()))
    assert out.cpu().tolist() == [768, 145, 1, 0]
    assert list(O.frame_lengths(cfg, m))[:3] == [768, 145, 1]


@pytest.mark.parametrize("B,T,Vv,U", [(3, 50, 32, 8), (2, 768, 32, 256), (4, 20, 7, 12), (2, 1499, 32, 256)])   # last: logits staged in two LDS chunks
def test_ctc_loss_and_grad(env, B, T, Vv, U):
    """tf.nn.ctc_loss semantics (losses.py:35-43): NLL vs the oracle (fp64), gradient vs torch autograd."""
    lib, torch, dev = env
    rng = np.random.default_rng(B * 1000 + T)
    logits = rng.normal(size=(B, T, Vv)).astype(np.float32) * 2
    labels = rng.integers(1, Vv, size=(B, U)).astype(np.int32)
    lab_len = rng.integers(0, min(U, T // 2) + 1, size=B).astype(np.int32)
    lab_len[0] = min(U, T // 2)
    for b in range(B):
        labels[b, lab_len[b]:] = 0
    log_len = np.full(B, T, np.int32)
    log_len[-1] = min(T, max(T - 3, 2 * int(lab_len[-1]) + 1))
    ref = O.ctc_nll(logits, labels, lab_len, log_len, blank=0)
    tl = dev_t(torch, dev, logits)
    nll = torch.empty((B,), device=dev)
    grad = torch.full((B, T, Vv), float("nan"), device=dev)
    N.check(lib.w2v2_ctc_loss(N.ptr(tl), B, T, Vv, N.ptr(dev_t(torch, dev, labels)), U, N.ptr(dev_t(torch, dev, lab_len)),
                              N.ptr(dev_t(torch, dev, log_len)), 0, N.ptr(nll), N.ptr(grad), stream()))
    got = nll.cpu().numpy()
    assert np.allclose(got, ref, rtol=2e-6, atol=1e-4), (got, ref)
    # gradient: torch CPU autograd on the same problem (fp64)
    lt = torch.from_numpy(logits).double().requires_grad_(True)
    lp = torch.log_softmax(lt, -1).transpose(0, 1)
    flat = torch.cat([torch.from_numpy(labels[b, :lab_len[b]].astype(np.int64)) for b in range(B)])
    loss = torch.nn.functional.ctc_loss(lp, flat, torch.from_numpy(log_len.astype(np.int64)),
                                        torch.from_numpy(lab_len.astype(np.int64)), blank=0, reduction="sum")
    loss.backward()
    g = grad.cpu().numpy()
    assert np.isfinite(g).all()
    assert H.max_err(g, lt.grad.numpy()) < 1e-5
    # nll-only call (grad = NULL) agrees
    nll2 = torch.empty((B,), device=dev)
    N.check(lib.w2v2_ctc_loss(N.ptr(tl), B, T, Vv, N.ptr(dev_t(torch, dev, labels)), U, N.ptr(dev_t(torch, dev, lab_len)),
                              N.ptr(dev_t(torch, dev, log_len)), 0, N.ptr(nll2), None, stream()))
    assert np.array_equal(nll2.cpu().numpy(), got)


@pytest.mark.parametrize("T,U,boost", [(768, 200, 25.0), (1499, 256, 40.0), (300, 100, -30.0)])
def test_ctc_blank_dominated_logits_need_more_than_fp64_range(env, T, U, boost):
    """The recursion runs on probabilities with a binary exponent per state pair (csrc/ctc.hip, round 6), not in log space.  What log
    space gave for free is range: with a blank-dominated model (early CTC training) the all-blank prefix outweighs the best label path
    by e^(U x boost) -- e^5000 at U = 200, boost 25: far beyond fp64 -- so a scheme with ONE scale per frame would flush every useful
    state to zero and report inf.  NLL against the log-space fp64 oracle, gradient against torch autograd; boost < 0 is the opposite
    corner (blank nearly impossible, the path forced through the labels)."""
    lib, torch, dev = env
    B, Vv = 2, 32
    rng = np.random.default_rng(T + U)
    logits = rng.normal(size=(B, T, Vv)).astype(np.float32) * 2
    logits[:, :, 0] += np.float32(boost)
    labels = rng.integers(1, Vv, size=(B, U)).astype(np.int32)
    lab_len = np.array([U, U // 2], np.int32)
    labels[1, lab_len[1]:] = 0
    log_len = np.array([T, T - 5], np.int32)
    ref = O.ctc_nll(logits, labels, lab_len, log_len, blank=0)
    assert np.isfinite(ref).all() and (boost < 0 or ref[0] > 700.0)        # (the premise: the loss is outside exp()'s fp64 range)
    tl = dev_t(torch, dev, logits)
    nll = torch.empty((B,), device=dev)
    grad = torch.full((B, T, Vv), float("nan"), device=dev)
    N.check(lib.w2v2_ctc_loss(N.ptr(tl), B, T, Vv, N.ptr(dev_t(torch, dev, labels)), U, N.ptr(dev_t(torch, dev, lab_len)),
                              N.ptr(dev_t(torch, dev, log_len)), 0, N.ptr(nll), N.ptr(grad), stream()))
    got = nll.cpu().numpy()
    assert np.allclose(got, ref, rtol=2e-6, atol=1e-3), (got, ref)
    lt = torch.from_numpy(logits).double().requires_grad_(True)
    lp = torch.log_softmax(lt, -1).transpose(0, 1)
    flat = torch.cat([torch.from_numpy(labels[b, :lab_len[b]].astype(np.int64)) for b in range(B)])
    loss = torch.nn.functional.ctc_loss(lp, flat, torch.from_numpy(log_len.astype(np.int64)), torch.from_numpy(lab_len.astype(np.int64)),
                                        blank=0, reduction="sum")
    loss.backward()
    g = grad.cpu().numpy()
    assert np.isfinite(g).all()
    assert H.max_err(g, lt.grad.numpy()) < 1e-5


def test_ctc_infeasible_is_inf(env):
    lib, torch, dev = env
    logits = np.zeros((1, 2, 5), np.float32)
    labels = np.array([[1, 1]], np.int32)          # "1 1" needs 3 frames
    nll = torch.empty((1,), device=dev)
    N.check(lib.w2v2_ctc_loss(N.ptr(dev_t(torch, dev, logits)), 1, 2, 5, N.ptr(dev_t(torch, dev, labels)), 2,
                              N.ptr(dev_t(torch, dev, np.array([2], np.int32))), N.ptr(dev_t(torch, dev, np.array([2], np.int32))),
                              0, N.ptr(nll), None, stream()))
    assert np.isinf(nll.cpu().numpy()[0])
