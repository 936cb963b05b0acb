// Shared pieces of the software-pipelined 128 x 256 GEMM kernels (gemm_bf16_sw.hip, gemm_split_sw.hip): LDS-DMA / counted-wait
// helpers and the LDS-staged epilogues of a wave that owns a 128 x 64 block of 32x32 accumulators (acc[4][2]).
#pragma once

#include "common.h"
#include "gemm_epilogue.h"

namespace w2v2 {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// Output stores of the LDS-staged epilogues with the non-temporal hint (tools-only knob W2V2_SW_NT).  Idea: C / C16 are consumed by a
// LATER kernel and never fit the 4-MB L2 of an XCD at B = 32, while the weight panel B16 (3.5-4.7 MB for the wide shapes) is re-read by
// every row tile -- plain stores allocate in L2 and evict it (PMC round 3: q|k|v fetched 218 MB for 41 MB of operands).  Measured in
// round 4 (profiles/r04_ab_sw_nt.txt, arms interleaved on one box): op level within +-0.7 % on every shape (q|k|v 91.6 vs 90.9 us,
// FFN up 141.7 vs 141.9), bf16 forward 10.88 vs 10.86-10.91 ms, fine-tune step 33.15-33.19 vs 33.22-33.24 ms -- no gain: the refetched
// B panel comes from the Infinity Cache fast enough to hide under the ring's ten-item prefetch.  Default: plain stores.
constexpr int SW_NT_DEFAULT = 0;
#ifdef W2V2_TUNING
#define SW_NT(g) ((g).nt != 0)
#else
#define SW_NT(g) (SW_NT_DEFAULT != 0)
#endif
template <typename V>
__device__ __forceinline__ void sw_store(V* p, const V& v, bool nt) {
    if (nt) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <int OFF>
__device__ __forceinline__ bf16x8 sw_read(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// one 1-KiB LDS-DMA piece: lanes fetch 16 B each from base + off (saddr form: scalar 64-bit base, 32-bit per-lane offset)
// into LDS bytes [dst, dst + 1024).  M0 is written in the statement that uses it (the compiler does not preserve it for asm);
// s_nop 0: one wait state between the SALU write of M0 and the LDS-DMA that reads it (the bases are SALU-computed from values
// made uniform at kernel entry, long before any DMA).
__device__ __forceinline__ void sw_dma(unsigned lds_dst, uint32_t off, const unsigned char* base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(off), "s"(base) : "memory");
}

template <int N>
__device__ __forceinline__ void sw_wait_vm() {
    if constexpr (N >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int V>
using IC = std::integral_constant<int, V>;

// ---- bf16-only epilogue through LDS (see gemm_bf16_pp.hip: column-major image, ds_write_b64, transposing reads, 16-byte stores)
__device__ __forceinline__ unsigned sw_cswz(int c) { return (unsigned)(((c & 3) << 2) | ((c >> 2) & 3)); }

// the wave's 128 x 64 bf16 image (column-major, written by ds_write_b64 as below) -> global rows of C16: transposing reads, 16-byte stores
__device__ __forceinline__ void sw_image_store(const bool nt, uint16_t* __restrict__ C16, int ldc, unsigned wbase, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int q = lane >> 4, l = lane & 15;
    const int cA = 8 * q + (l >> 2);
    const unsigned rdA = wbase + (unsigned)cA * 256u, preA = ((unsigned)(l & 3) ^ sw_cswz(cA)) << 3;
    const unsigned rdB = rdA + 4u * 256u, preB = ((unsigned)(l & 3) ^ sw_cswz(cA + 4)) << 3;
    u32x2 lo[16], hi[16];
    uint16_t* const dst = C16 + (int64_t)l * ldc + 8 * q;
#define SW_RD4(G)                                                                                                            \
    _Pragma("unroll") for (int t = 4 * (G); t < 4 * (G) + 4; ++t) {                                                          \
        const unsigned R = (unsigned)(t >> 1), Hc = (unsigned)(t & 1);                                                       \
        const unsigned a1 = rdA + Hc * 8192u + ((32u * R) ^ preA), a2 = rdB + Hc * 8192u + ((32u * R) ^ preB);               \
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[t]) : "v"(a1));                                                   \
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi[t]) : "v"(a2));                                                   \
    }
#define SW_ST4(G, WAIT)                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(%8)"                                                                                     \
                 : "+v"(lo[4 * (G)]), "+v"(hi[4 * (G)]), "+v"(lo[4 * (G) + 1]), "+v"(hi[4 * (G) + 1]), "+v"(lo[4 * (G) + 2]),   \
                   "+v"(hi[4 * (G) + 2]), "+v"(lo[4 * (G) + 3]), "+v"(hi[4 * (G) + 3])                                       \
                 : "n"(WAIT));                                                                                               \
    _Pragma("unroll") for (int t = 4 * (G); t < 4 * (G) + 4; ++t) {                                                          \
        u32x4 o;                                                                                                             \
        o[0] = lo[t][0]; o[1] = lo[t][1]; o[2] = hi[t][0]; o[3] = hi[t][1];                                                  \
        sw_store(reinterpret_cast<u32x4*>(dst + (int64_t)(16 * (t >> 1)) * ldc + 32 * (t & 1)), o, nt);                      \
    }
    SW_RD4(0)
    SW_RD4(1)
    SW_ST4(0, 8)
    SW_RD4(2)
    SW_ST4(1, 8)
    SW_RD4(3)
    SW_ST4(2, 8)
    SW_ST4(3, 0)
#undef SW_RD4
#undef SW_ST4
}

template <int ACT>
__device__ __forceinline__ void sw_epilogue_bf16(const bool nt, const f32x16 (&acc)[4][2], uint16_t* __restrict__ C16, const float* __restrict__ bias, int ldc,
                                                 unsigned wbase, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const unsigned pre_w = ((sw_cswz(li) ^ (unsigned)lh) << 3);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const float bv = bias ? bias[nt * 32 + li] : 0.0f;
        const unsigned colbase = wbase + (unsigned)(nt * 32 + li) * 256u;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x16 v = acc[mt][nt];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += bv;
            if constexpr (ACT == 1) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2_t t = gelu_erf_fast2(f32x2_t{v[r], v[r + 1]});
                    v[r] = t[0];
                    v[r + 1] = t[1];
                }
            } else if constexpr (ACT == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = gelu_tanh(v[r]);
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                u32x2 w;
                w[0] = pack_bf16_rne(v[4 * gq], v[4 * gq + 1]);
                w[1] = pack_bf16_rne(v[4 * gq + 2], v[4 * gq + 3]);
                const unsigned a = colbase + ((unsigned)((mt * 8 + 2 * gq) << 3) ^ pre_w);
                asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(w) : "memory");
            }
        }
    }
    sw_image_store(nt, C16, ldc, wbase, lane);
}

// ---- fp32 epilogue through LDS (outputs that stay fp32: the residual stream, the data gradients).  From registers a lane writes
// 128 single dwords (2 rows x 128 B per store instruction) and reads the residual the same way; here the wave's tile goes through its
// 16 KiB of LDS in two halves of 64 rows as a row-major fp32 image (ds_write_b32: 32 consecutive dwords per half-wave), comes back
// as 16 bytes of one row per lane (ds_read_b128, 4 whole 256-byte rows per instruction), and residual loads, fp32 stores (16 B per
// lane) and the optional bf16 shadow (8 B per lane) all move whole rows.  Same element arithmetic in the same order as
// gemm_epilogue: (acc + bias) -> act -> + residual.
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int ACT, bool FASTG = true>
__device__ __forceinline__ void sw_epilogue_f32(const bool nt, const f32x16 (&acc)[4][2], float* __restrict__ C, uint16_t* __restrict__ C16,
                                                const float* __restrict__ R, const float* __restrict__ bias, int ldc, unsigned wbase, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const unsigned wr0 = wbase + (unsigned)(4 * lh) * 256u + (unsigned)li * 4u;      // register r adds ((r & 3) + 8 (r >> 2)) rows
    const int row_l = lane >> 4, ch = lane & 15;                                     // read side: row 4 t + row_l, 16-byte chunk ch
    const unsigned rd0 = wbase + (unsigned)row_l * 256u + (unsigned)ch * 16u;
    float bv[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bv[nt] = bias ? bias[nt * 32 + li] : 0.0f;
    // The residual rows of a half (16 loads of 16 bytes per lane) are requested BEFORE that half's trip through LDS -- those of the
    // second half while the first half's rows are still being read back -- so their HBM latency runs under the LDS traffic.  (First
    // version: four loads at a time beside the four LDS reads that needed them; a phase trace of the FFN down-projection showed the
    // epilogue at 27 k cycles with the block alone on its CU: eight exposed round trips.)  Asm loads (scalar base + one 32-bit lane
    // offset) so that they stay where they are written, retired by counted waits: vmcnt is in order on gfx950, and what may still be
    // in flight behind the four loads a group needs are the later loads already requested and at least one store per row group
    // already written.  Order: 8 loads | first half's LDS writes | 8 loads | row groups 0, 1 | the second half's 16 loads | groups 2, 3
    // | second half's LDS writes | its four groups -> 12, 12, 28, 28 requests may stay in flight for the first half's groups, 20 for
    // each of the second's.  (More loads up front made the allocator spill load destinations -- which the compiler then stores
    // before they have landed; tools/scratch_scan.sh guards the instance against that.)
    f32x4 rr[2][16];
    const unsigned rrow = ((unsigned)row_l * (unsigned)ldc + 4u * (unsigned)ch) * 4u;     // byte offset inside the wave's tile (< 2^32: ldc < 2^23)
    const uint64_t rbits = reinterpret_cast<uint64_t>(R);
    const char* const Rs = reinterpret_cast<const char*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(rbits >> 32)) << 32) |
                                                         (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)rbits));   // (R is wave-uniform; the builtin returns int: no sign extension into the high word)
#define SW_F32_RES(H, T0, T1)                                                                                                     \
    if (R) {                                                                                                                      \
        _Pragma("unroll") for (int t = (T0); t < (T1); ++t)                                                                       \
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rr[H][t]) : "v"(rrow), "s"(Rs + (size_t)(64 * (H) + 4 * t) * (size_t)ldc * 4u) : "memory"); \
    }
    asm volatile("" : "+v"(bv[0]), "+v"(bv[1]));      // (the bias has landed before the counted loads start: the compiler's own wait for it would drain them)
    constexpr int EARLY = ACT == 0 ? 8 : 0;      // (an activation's temporaries leave no room for loads in flight next to all 128 accumulators)
    SW_F32_RES(0, 0, EARLY)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // (the first half's reads have retired)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                f32x16 v = acc[2 * h + mh][nt];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += bv[nt];
                if constexpr (ACT == 1 && !FASTG) {      // (the fp32-grade split GEMM: erff, as the fp32 path)
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = gelu_erf(v[r]);
                } else if constexpr (ACT == 1) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2_t t = gelu_erf_fast2(f32x2_t{v[r], v[r + 1]});
                        v[r] = t[0];
                        v[r + 1] = t[1];
                    }
                } else if constexpr (ACT == 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = gelu_tanh(v[r]);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // (the row / column-half displacement as the instruction's offset field: one address register for all 128 writes)
                    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(wr0), "v"(v[r]), "n"((mh * 32 + (r & 3) + 8 * (r >> 2)) * 256 + nt * 128) : "memory");
                }
            }
        if (h == 0) {
            SW_F32_RES(0, EARLY, 16)
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int64_t rbase = (int64_t)(64 * h + row_l) * ldc + 4 * ch;
        // 16 reads of 4 rows each, in groups of four
#define SW_F32_GROUP(TG)                                                                                                          \
    {                                                                                                                             \
        f32x4 x0, x1, x2, x3;                                                                                                     \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x0) : "v"(rd0), "n"((4 * (TG) + 0) * 1024));                          \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x1) : "v"(rd0), "n"((4 * (TG) + 1) * 1024));                          \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x2) : "v"(rd0), "n"((4 * (TG) + 2) * 1024));                          \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x3) : "v"(rd0), "n"((4 * (TG) + 3) * 1024));                          \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));                                            \
        if (R) {                                                                                                                  \
            if (h == 0 && (TG) < 2)                                                                                               \
                asm volatile("s_waitcnt vmcnt(12)" : "+v"(rr[h][4 * (TG) + 0]), "+v"(rr[h][4 * (TG) + 1]), "+v"(rr[h][4 * (TG) + 2]), "+v"(rr[h][4 * (TG) + 3]) : : "memory"); \
            else if (h == 0)                                                                                                      \
                asm volatile("s_waitcnt vmcnt(28)" : "+v"(rr[h][4 * (TG) + 0]), "+v"(rr[h][4 * (TG) + 1]), "+v"(rr[h][4 * (TG) + 2]), "+v"(rr[h][4 * (TG) + 3]) : : "memory"); \
            else                                                                                                                  \
                asm volatile("s_waitcnt vmcnt(20)" : "+v"(rr[h][4 * (TG) + 0]), "+v"(rr[h][4 * (TG) + 1]), "+v"(rr[h][4 * (TG) + 2]), "+v"(rr[h][4 * (TG) + 3]) : : "memory"); \
        }                                                                                                                         \
        SW_F32_OUT(x0, 4 * (TG) + 0)                                                                                              \
        SW_F32_OUT(x1, 4 * (TG) + 1)                                                                                              \
        SW_F32_OUT(x2, 4 * (TG) + 2)                                                                                              \
        SW_F32_OUT(x3, 4 * (TG) + 3)                                                                                              \
    }
#define SW_F32_OUT(X, T)                                                                                                          \
    {                                                                                                                             \
        const f32x4 o = R ? (X) + rr[h][T] : (X);                                                                                 \
        const int64_t off = rbase + (int64_t)(4 * (T)) * ldc;                                                                     \
        if (C) sw_store(reinterpret_cast<f32x4*>(C + off), o, nt);                                                                \
        if (C16) {                                                                                                                \
            u32x2 pk;                                                                                                             \
            pk[0] = pack_bf16_rne(o[0], o[1]);                                                                                    \
            pk[1] = pack_bf16_rne(o[2], o[3]);                                                                                    \
            sw_store(reinterpret_cast<u32x2*>(C16 + off), pk, nt);                                                                \
        }                                                                                                                         \
    }
        SW_F32_GROUP(0)
        SW_F32_GROUP(1)
        if (h == 0) {
            SW_F32_RES(1, 0, 16)
            __builtin_amdgcn_sched_barrier(0);
        }
        SW_F32_GROUP(2)
        SW_F32_GROUP(3)
#undef SW_F32_GROUP
#undef SW_F32_OUT
    }
#undef SW_F32_RES
}

// ---- plane epilogue (precision modes bf16x3 / f16x2: the consumer GEMM streams its activation as three bf16 planes whose sum is the
// fp32 value EXACTLY, or as the two fp16 terms of value x F16X2_ACT_SCALE -- gemm_split_sw.hip; that kernel applies the activation
// before it calls this).  acc + bias in place, then plane p = round(rest), rest -= plane p (exact in fp32), each plane through the
// wave's LDS image like sw_epilogue_bf16 (the transposing read moves 16-bit elements: fp16 and bf16 alike).  An LDS pipe executes one wave's
// operations in order and the image is the wave's own, so a pass's writes cannot overtake the previous pass's reads.
template <int FMT>
__device__ __forceinline__ void sw_epilogue_planes(const bool nt, f32x16 (&acc)[4][2], uint16_t* __restrict__ C16, int64_t plane, const float* __restrict__ bias,
                                                   int ldc, unsigned wbase, int lane, int* range_flag) {
    const int li = lane & 31, lh = lane >> 5;
    const unsigned pre_w = ((sw_cswz(li) ^ (unsigned)lh) << 3);
    if (bias) {
#pragma unroll
        for (int nt_ = 0; nt_ < 2; ++nt_) {
            const float bv = bias[nt_ * 32 + li];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt_][r] += bv;
        }
    }
    if constexpr (FMT == PF_F16X2) {      // x S, saturated to fp16's range (a saturated value is wrong: reported through the sticky flag)
        bool ovf = false;
#pragma unroll
        for (int nt_ = 0; nt_ < 2; ++nt_)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float t = acc[mt][nt_][r] * F16X2_ACT_SCALE;
                    ovf |= !(fabsf(t) <= F16X2_MAX);
                    acc[mt][nt_][r] = __builtin_amdgcn_fmed3f(t, -F16X2_MAX, F16X2_MAX);
                }
        report_overflow(range_flag, ovf);
    }
#pragma unroll
    for (int p = 0; p < plane_count(FMT); ++p) {
#pragma unroll
        for (int nt_ = 0; nt_ < 2; ++nt_) {
            const unsigned colbase = wbase + (unsigned)(nt_ * 32 + li) * 256u;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    u32x2 w;
                    if constexpr (FMT == PF_F16X2) {
                        const h2_t a = {(_Float16)acc[mt][nt_][4 * gq], (_Float16)acc[mt][nt_][4 * gq + 1]};
                        const h2_t b = {(_Float16)acc[mt][nt_][4 * gq + 2], (_Float16)acc[mt][nt_][4 * gq + 3]};
                        w[0] = __builtin_bit_cast(unsigned, a);
                        w[1] = __builtin_bit_cast(unsigned, b);
                        if (p == 0) {
                            acc[mt][nt_][4 * gq] -= (float)a[0];
                            acc[mt][nt_][4 * gq + 1] -= (float)a[1];
                            acc[mt][nt_][4 * gq + 2] -= (float)b[0];
                            acc[mt][nt_][4 * gq + 3] -= (float)b[1];
                        }
                    } else {
                        w[0] = pack_bf16_rne(acc[mt][nt_][4 * gq], acc[mt][nt_][4 * gq + 1]);
                        w[1] = pack_bf16_rne(acc[mt][nt_][4 * gq + 2], acc[mt][nt_][4 * gq + 3]);
                        if (p < 2) {
                            acc[mt][nt_][4 * gq] -= __uint_as_float(w[0] << 16);
                            acc[mt][nt_][4 * gq + 1] -= __uint_as_float(w[0] & 0xffff0000u);
                            acc[mt][nt_][4 * gq + 2] -= __uint_as_float(w[1] << 16);
                            acc[mt][nt_][4 * gq + 3] -= __uint_as_float(w[1] & 0xffff0000u);
                        }
                    }
                    const unsigned a_ = colbase + ((unsigned)((mt * 8 + 2 * gq) << 3) ^ pre_w);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(a_), "v"(w) : "memory");
                }
            }
        }
        sw_image_store(nt, C16 + (int64_t)p * plane, ldc, wbase, lane);
    }
}

}  // namespace

}  // namespace w2v2
