#!/usr/bin/env python
"""Per-block phase trace of the shadow-fed bf16 GEMM (tools-only build: `python gsoc-wav2vec2_amd/build.py --tuning`).

    W2V2_NATIVE_LIB=gsoc-wav2vec2_amd/lib/libw2v2_tuning.so python tools/gemm16_trace.py [shape ...]

For each base-model shape at B = 32 it launches the GEMM with the operands / outputs the forward uses (q|k|v: bf16 output only;
out-projection / FFN down: fp32 output + residual; FFN up: GELU, bf16 only), once timed without the trace and once with it, and
prints where a block's cycles go: entry -> first tile landed, every k step, last MFMA block, epilogue until its stores retire;
plus how many blocks a CU ran and the spread of their start times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd"))
os.environ.setdefault("W2V2_NATIVE_LIB", os.path.join(ROOT, "gsoc-wav2vec2_amd", "lib", "libw2v2_tuning.so"))
import ctypes as C
import numpy as np
import torch
from wav2vec2 import _native as N

lib = N.load()
P, I32, I64 = C.c_void_p, C.c_int32, C.c_int64
lib.w2v2_tune_set_trace.restype = C.c_int
lib.w2v2_tune_set_trace.argtypes = [P]
dev = torch.device("cuda:0")
B = int(os.environ.get("TRACE_BATCH", 32)); BT = B * 768
# name: M, N, K, lda, strideA, nbatch, act, fp32 out, bf16 out, residual
SHAPES = {"qkv": (BT, 2304, 768, 768, 0, 1, 0, False, True, False), "out": (BT, 768, 768, 768, 0, 1, 0, True, False, True),
          "ffn1": (BT, 3072, 768, 768, 0, 1, 1, False, True, False), "ffn2": (BT, 768, 3072, 3072, 0, 1, 0, True, False, True),
          "conv1": (24599, 512, 1536, 1024, 49199 * 512, B, 1, False, True, False),
          "conv4": (3074, 512, 1536, 1024, 6149 * 512, B, 1, False, True, False)}
VARIANT = int(os.environ.get("TRACE_VARIANT", 0))      # 0 by shape, 1 = 128 x 128, 2 = 128 x 256 software-pipelined
names = sys.argv[1:] or ["qkv", "out", "ffn1", "ffn2", "conv1"]
for name in names:
    M, Nn, K, lda, sA, nb, act, f32o, b16o, res = SHAPES[name]
    a_elems = (nb - 1) * sA + (M - 1) * lda + K if sA else M * lda
    A16 = torch.randn(a_elems, device=dev).to(torch.bfloat16)
    B16 = (torch.randn(Nn, K, device=dev) * 0.05).to(torch.bfloat16)
    Cf = torch.empty(nb * M * Nn, device=dev) if f32o else None
    Ch = torch.empty(nb * M * Nn, device=dev, dtype=torch.bfloat16) if b16o else None
    bias = torch.randn(Nn, device=dev)
    R = torch.randn(nb * M * Nn, device=dev) if res else None
    st = N.current_stream()

    def run():
        N.check(lib.w2v2_op_gemm_bf16_shadows(N.ptr(A16), lda, sA, N.ptr(B16), N.ptr(Cf), N.ptr(Ch), Nn, M * Nn, N.ptr(bias), N.ptr(R), M, Nn, K, nb, act, VARIANT, st))
    lib.w2v2_tune_set_trace(None)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tiles = ((M + 127) // 128) * ((Nn + 127) // 128) * nb
    tr = torch.zeros(tiles * 32, dtype=torch.int64, device=dev)
    lib.w2v2_tune_set_trace(tr.data_ptr())
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ms_tr = e0.elapsed_time(e1)
    lib.w2v2_tune_set_trace(None)
    t = tr.cpu().numpy().reshape(tiles, 32)
    cnt = int(t[0, 31]); nk = K // 64
    if cnt == 8:          # gemm_bf16_sw_kernel: entry, prologue done, steady loop done, tail done + barrier, epilogue issued, stores retired
        tiles = ((M + 127) // 128) * (Nn // 256) * nb
        t = t[:tiles]
        clk = t[:, 2:8].astype(np.float64); d = np.diff(clk, axis=1)
        q = lambda v: "%7.0f %7.0f %7.0f" % tuple(np.percentile(v, [10, 50, 90]))
        print(f"== {name} [128x256 software-pipelined, 2 blocks / CU]: {ms:.4f} ms = {2.0 * M * Nn * K * nb / ms / 1e9:.0f} TF; {tiles} tiles, {nk} K tiles")
        for lab, col in (("entry -> A_0, B_0 of K tile 0 in registers", 0), (f"steady loop ({nk - 2} K tiles)", 1), ("last two K tiles + barrier", 2),
                         ("epilogue until stores issued", 3), ("stores retired (vmcnt 0)", 4)):
            print(f"   {lab:44s} {q(d[:, col])}" + (f"   per K tile {q(d[:, col] / (nk - 2))} (1024 = the pipe to itself)" if col == 1 else ""))
        print(f"   total {q(clk[:, -1] - clk[:, 0])}")
        hw = t[:, 0]
        cu_key = ((hw >> 32) << 16) | ((hw & 0xFFFFFFFF) >> 8 & 0xF) | (((hw & 0xFFFFFFFF) >> 13 & 0x7) << 4) | (((hw & 0xFFFFFFFF) >> 12 & 1) << 7)
        uniq, per_cu = np.unique(cu_key, return_counts=True)
        print(f"   {len(uniq)} distinct CU ids; blocks per CU min / median / max {per_cu.min()} / {int(np.median(per_cu))} / {per_cu.max()}")
        continue
    if cnt == 10:         # gemm_bf16_pp_kernel with the LDS-staged bf16 epilogue: + barrier passed, LDS written, stores issued
        tiles = ((M + 255) // 256) * ((Nn + 255) // 256) * nb
        t = t[:tiles]
        clk = t[:, 2:10].astype(np.float64); d = np.diff(clk, axis=1)
        q = lambda v: "%7.0f %7.0f %7.0f" % tuple(np.percentile(v, [10, 50, 90]))
        print(f"== {name} [256x256 ping-pong, LDS-staged epilogue]: {ms:.4f} ms = {2.0 * M * Nn * K * nb / ms / 1e9:.0f} TF; {tiles} tiles, {nk} K tiles")
        for lab, col in (("entry -> first half-tiles landed", 0), (f"steady loop ({nk - 2} K tiles)", 1), ("last two K tiles", 2), ("final barrier", 3),
                         ("bias / act / cvt + LDS writes", 4), ("tr reads + store issue", 5), ("stores retired (vmcnt 0)", 6)):
            print(f"   {lab:36s} {q(d[:, col])}")
        print(f"   total {q(clk[:, -1] - clk[:, 0])}")
        continue
    if cnt == 7:          # gemm_bf16_pp_kernel: entry, items 0-1 landed, end of the steady loop, end of the last two K tiles, stores retired
        tiles = ((M + 255) // 256) * ((Nn + 255) // 256) * nb
        t = t[:tiles]
        clk = t[:, 2:7].astype(np.float64); d = np.diff(clk, axis=1)
        q = lambda v: "%7.0f %7.0f %7.0f" % tuple(np.percentile(v, [10, 50, 90]))
        print(f"== {name} [256x256 ping-pong]: M={M} N={Nn} K={K} batch={nb}: {ms:.4f} ms = {2.0 * M * Nn * K * nb / ms / 1e9:.0f} TF (traced launch {ms_tr:.4f} ms); {tiles} tiles, {nk} K tiles")
        print(f"   cycles per block (p10 / p50 / p90):  total {q(clk[:, -1] - clk[:, 0])}")
        print(f"   entry -> first two half-tiles landed {q(d[:, 0])}")
        print(f"   steady loop ({nk - 2} K tiles)           {q(d[:, 1])}    per K tile {q(d[:, 1] / max(1, nk - 2))}   (ideal 2048)")
        print(f"   last two K tiles                    {q(d[:, 2])}")
        print(f"   epilogue until stores retired       {q(d[:, 3])}")
        hw = t[:, 0]
        cu_key = ((hw >> 32) << 16) | ((hw & 0xFFFFFFFF) >> 8 & 0xF) | (((hw & 0xFFFFFFFF) >> 13 & 0x7) << 4) | (((hw & 0xFFFFFFFF) >> 12 & 1) << 7)
        uniq, per_cu = np.unique(cu_key, return_counts=True)
        wall = t[:, 1].astype(np.float64) * 10.0
        print(f"   {len(uniq)} distinct CU ids; blocks per CU min / median / max {per_cu.min()} / {int(np.median(per_cu))} / {per_cu.max()}; block starts span {(wall.max() - wall.min()) / 1e3:.1f} us")
        continue
    assert cnt == 2 + 1 + 1 + (nk - 1) + 1 + 1, (cnt, nk)
    clk = t[:, 2:cnt].astype(np.float64)
    d = np.diff(clk, axis=1)            # [prologue, k-step 0 .. nk-2, last compute, epilogue]
    q = lambda v: "%7.0f %7.0f %7.0f" % tuple(np.percentile(v, [10, 50, 90]))
    total = clk[:, -1] - clk[:, 0]
    print(f"== {name}: M={M} N={Nn} K={K} batch={nb}: {ms:.4f} ms = {2.0 * M * Nn * K * nb / ms / 1e9:.0f} TF (traced launch {ms_tr:.4f} ms); {tiles} tiles, {nk} k-steps")
    print(f"   cycles per block (p10 / p50 / p90):  total {q(total)}")
    print(f"   entry -> first tile landed          {q(d[:, 0])}")
    steps = d[:, 1:nk]
    print(f"   one k step (all steps pooled)       {q(steps.reshape(-1))}    sum over the loop {q(steps.sum(axis=1))}")
    print(f"   per-step medians: " + " ".join("%.0f" % v for v in np.median(steps, axis=0)))
    print(f"   last MFMA block                     {q(d[:, nk])}")
    print(f"   epilogue until stores retired       {q(d[:, nk + 1])}")
    hw = t[:, 0]
    cu_key = ((hw >> 32) << 16) | ((hw & 0xFFFFFFFF) >> 8 & 0xF) | (((hw & 0xFFFFFFFF) >> 13 & 0x7) << 4) | (((hw & 0xFFFFFFFF) >> 12 & 1) << 7)
    uniq, per_cu = np.unique(cu_key, return_counts=True)
    wall = t[:, 1].astype(np.float64) * 10.0      # ns (100 MHz)
    print(f"   {len(uniq)} distinct (xcc, se, sh, cu) ids; blocks per CU min / median / max {per_cu.min()} / {int(np.median(per_cu))} / {per_cu.max()}; "
          f"block starts span {(wall.max() - wall.min()) / 1e3:.1f} us; ideal MFMA cycles per tile {nk * 8 * 32 * 2}")
