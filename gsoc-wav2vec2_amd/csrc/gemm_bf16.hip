// Mixed-precision GEMM for the bf16 configurations (BASELINE configs[2] and [4]):
// fp32 operands in HBM, rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) while they are
// staged into LDS, multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; bias, GELU,
// residual and the store stay fp32.  Numerically: C = act(bf16(A) . bf16(B) + bias) + residual with
// exact products and an fp32 running sum -- what a bf16 autocast Dense/Conv1D computes.
//
// Same contract as gemm_f32.hip (overlapping-row A for the strided convs, per-batch strides for the
// split-K weight gradients), so the whole model -- forward and backward -- switches precision by
// routing launch_gemm here (gemm_set_precision).
//
// Tile: BM x 128 x 64, (BM/64) x 2 waves, each wave a 64x64 sub-tile = 2x2 accumulators of
// 32x32.  One K tile is 16 MFMAs per wave (512 matrix-pipe cycles) against 16 ds_read_b128.
// LDS image: rows of 64 bf16 = 128 B (A rows = m, B rows = n, both k-contiguous so a fragment is ONE
// 16-byte read of 8 consecutive k); the 16-B slot index is XOR-swizzled per row (swz) so that the
// fragment reads, the A stores and the transposing B stores are all bank-conflict free.  B arrives
// n-contiguous from HBM ([K, N] TF kernel layout); the transpose to k-contiguous happens in registers:
// a thread owns an 8(k) x 4(n) patch, loads it as eight float4 and writes four 16-byte rows.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"

namespace w2v2 {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int BK = 64;          // bf16 elements per K tile = one 128-byte LDS row
constexpr int ROWB = 128;       // LDS row bytes

struct Gemm16Args {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* residual;
    int64_t lda, ldb, ldc, strideA, strideB, strideC;
    int M, N, K, act;
    int tiles_m, tiles_n;
    // optional bf16 shadows (GemmShadows): A16 has A's shape and strides, B16 is B transposed ([N][K], ld = ldb16)
    const uint16_t* A16;
    const uint16_t* B16;
    uint16_t* C16;
    int64_t ldb16;
    const uint16_t* B16p;      // weight-gradient tr form: plain (K, N) bf16 copy of B (row stride ldb, batch stride strideB)
    // two-level batch (grouped conv as GEMM): z = zo * zmod + zi.  A advances with z; B16 and bias with zi; C / residual
    // with zo * strideC2 + zi * strideC.  zmod = 0: plain batch (C advances with z * strideC, B16 and bias are shared).
    int zmod;
    int64_t strideB16, strideC2, strideBias, strideB2;
    // SRC 7 (weight gradient dW = X^T dY): optional column sums of B over this batch's K rows -> colsum[z * strideCS + n]
    // (the bias gradient: B = dY is already in registers while it is staged, so the sums cost 8 adds per patch column)
    float* colsum;
    int64_t strideCS;
    int kseg;              // SRC 7: K as segments of kseg rows, segA / segB elements apart (GemmShadows::kseg); 0 = contiguous
    int64_t segA, segB;
    int64_t validK;        // tr form: rows of the K dimension that exist, over all batches (batch z owns [z K, (z + 1) K)); rows beyond
                           // read as zero, so neither the row count nor its split into slabs has to be a multiple of the K tile
#ifdef W2V2_TUNING
    unsigned long long* trace = nullptr;      // tools-only build: per-block phase stamps (tools/gemm16_trace.py), 32 words per block
    int abl = 0;  // timing ablations of the tools-only build (W2V2_GEMM16_ABL, results are wrong by construction): 1 = no operand
                  // traffic in the K loop, 2 = no MFMAs, 4 = no epilogue.  The shipping kernel has no such field and no such branches.
#endif
};

// two fp32 -> one dword of two bf16, round to nearest even (gfx950 instruction; no builtin in ROCm 7.2)
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// 16-byte-slot swizzle of an LDS row (8 slots per 128-byte row).  Chosen so that all three access patterns are
// bank-conflict free: ds_read_b128 fragment reads (16-lane groups {0-3,12-15,20-27}, ... of consecutive rows),
// the A stores (16 lanes = one row) and the transposing B stores (8 lanes = rows 4 q + j or 2 q + j, q = 8g..8g+7).
__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 1); }

template <int PN> struct FVec;
template <> struct FVec<4> { using type = f32x4; };
template <> struct FVec<2> { using type = f32x2; };

// SRC: where the operands come from.  0 = fp32 with scalar guards (any shape), 1 = fp32, 16-byte loads,
// 2 = A from its bf16 shadow, 3 = B from its bf16 [N][K] shadow, 4 = both shadows (no conversion at all: the tile
// step streams 32 KiB instead of 64), 5 = both shadows copied HBM/L2 -> LDS by global_load_lds_dwordx4 (the swizzle
// then goes on the per-lane SOURCE address, as in gemm_f32.hip), 7 = fp32 with A TRANSPOSED in memory ((K, M), the
// activation itself in a weight-gradient GEMM  dW = X^T dY): A takes B's register-transposing path, no transposed copy.  Shadows hold exactly the values the fp32 path would round to, so all five
// produce bit-identical results.
template <int SRC, int BM, int BN, int WM, int WN, int MINB>
__global__ __launch_bounds__(WM* WN * 64, MINB) void gemm_bf16_kernel(Gemm16Args g) {
    constexpr bool FAST = SRC >= 1, A16 = SRC == 2 || (SRC >= 4 && SRC <= 6), B16 = SRC == 3 || (SRC >= 4 && SRC <= 6);
    constexpr bool DMA = SRC == 5 || SRC == 6;      // both shadows, LDS-DMA staging (no registers, no ds_write)
    constexpr bool AT = SRC == 7;                   // A given TRANSPOSED ((K, M) fp32, m-contiguous): staged like B
    constexpr int NS = SRC == 6 ? 4 : 2;   // LDS stages; SRC 6 = 4-stage ring, three tiles in flight across raw barriers
    constexpr int NT = WM * WN * 64;
    constexpr int NA16 = BM * 8 / NT, NB16 = BN * 8 / NT;   // 16-byte (8 x bf16) chunks per thread when a shadow is the source
    constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NTL = WTN / 32;   // wave tile, 32x32 accumulators
    constexpr int NA = BM * 16 / NT;              // float4 chunks of the A tile per thread
    constexpr int PN = (2 * BN >= NT) ? 4 : 2;    // B patch = 8(k) x PN(n) per thread, loaded as 8 float4 | float2
    constexpr int NQ = BN / PN;                   // patches across n
    constexpr int NB = 8 * NQ / NT;               // patches per thread
    constexpr int STAGE = (BM + BN) * ROWB;       // bytes per LDS stage
    static_assert(NA >= 1 && NB >= 1 && MT >= 1 && NTL >= 1, "bad tile / wave grid");
    using bvec = typename FVec<PN>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform: LDS-DMA destinations and piece indices become scalar)
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;

    // XCD-aware tile order (see gemm_f32.hip): each XCD walks a contiguous run of tiles, N fastest
    const int nwg = g.tiles_m * g.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / g.tiles_n, tn = bid % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.z;
    const float* __restrict__ A = g.A + (int64_t)z * g.strideA;
    const float* __restrict__ Bm = g.B + (g.zmod ? (int64_t)(z / g.zmod) * g.strideB2 + (int64_t)(z % g.zmod) * g.strideB
                                                 : (int64_t)z * g.strideB);

#ifdef W2V2_TUNING
    const int abl = g.abl;
    unsigned long long* const trc = g.trace ? g.trace + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 32 : nullptr;
    int trc_n = 2;
    if (trc && tid == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trc[0] = ((unsigned long long)xcc << 32) | hwid;
        trc[1] = wall_clock64();
        trc[trc_n++] = clock64();
    }
#define W2V2_TRC() do { if (trc && tid == 0 && trc_n < 31) trc[trc_n++] = clock64(); } while (0)
#else
    constexpr int abl = 0;      // (the conditions below fold away: the DMA issue is unconditional in the shipping kernel)
#define W2V2_TRC() do { } while (0)
#endif
    const int nk = (g.K + BK - 1) / BK;
    const bool do_colsum = AT && g.colsum != nullptr && tm == 0;      // block-uniform

    // ---- global -> register staging.  Every wave-level load is fully coalesced: A as 16 lanes x 16 B per
    // 256-byte row, B as NQ lanes x (4 PN) B per k-row.  The vector L1 (64 B/clk/CU) is the resource this kernel
    // leans on hardest -- 64 KiB of fp32 per 128x128x64 tile step -- so no request may touch a line twice.
    f32x4 ra[NA];
    bvec rb[NB][8];
    float cs[AT ? NB : 1][PN];        // per-thread partial column sums of B (SRC 7 with g.colsum, row-tile 0 only)
#pragma unroll
    for (int i = 0; i < (AT ? NB : 1); ++i)
#pragma unroll
        for (int j = 0; j < PN; ++j) cs[i][j] = 0.f;
    int64_t a_off[NA], b_off[NB];
    int a_lds[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int idx = tid + i * NT, r = idx >> 4, sl = idx & 15;     // sl: 16-byte slot of the fp32 row = 4 k
        int row = m0 + r;
        row = row < g.M ? row : g.M - 1;              // clamped rows feed accumulators never stored
        a_off[i] = (int64_t)row * g.lda + sl * 4;
        a_lds[i] = r * ROWB + (((sl >> 1) ^ swz(r)) << 4) + (sl & 1) * 8;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idx = tid + i * NT, q = idx % NQ, ks = idx / NQ;
        int col = n0 + PN * q;
        col = col < g.N ? col : (g.N >= PN ? g.N - PN : 0);
        b_off[i] = (int64_t)(ks * 8) * g.ldb + col;
    }

    // A^T source (SRC 7): 8(k) x 4(m) patches exactly like B; element (m, k) of A lives at A[k * lda + m]
    constexpr int NQA = BM / 4, NAT = 8 * NQA / NT;
    f32x4 rat[AT ? NAT : 1][8];
    int64_t at_off[AT ? NAT : 1];
    if constexpr (AT) {
#pragma unroll
        for (int i = 0; i < NAT; ++i) {
            const int idx = tid + i * NT, q = idx % NQA, ks = idx / NQA;
            int row = m0 + 4 * q;
            row = row < g.M ? row : (g.M >= 4 ? g.M - 4 : 0);
            at_off[i] = (int64_t)(ks * 8) * g.lda + row;
        }
    }

    // shadow sources: 8 lanes x 16 B = one 128-byte tile row, stored to LDS as loaded
    u32x4 ra16[A16 ? NA16 : 1], rb16[B16 ? NB16 : 1];
    const uint16_t* a16_src[A16 ? NA16 : 1];
    const uint16_t* b16_src[B16 ? NB16 : 1];
    int a16_lds[A16 ? NA16 : 1], b16_lds[B16 ? NB16 : 1];
    if constexpr (A16) {
        const uint16_t* A16p = g.A16 + (int64_t)z * g.strideA;
#pragma unroll
        for (int i = 0; i < NA16; ++i) {
            const int idx = tid + i * NT, r = idx >> 3, ks = idx & 7;
            int row = m0 + r;
            row = row < g.M ? row : g.M - 1;
            a16_src[i] = A16p + (int64_t)row * g.lda + ks * 8;
            a16_lds[i] = r * ROWB + ((ks ^ swz(r)) << 4);
        }
    }
    if constexpr (B16) {
#pragma unroll
        for (int i = 0; i < NB16; ++i) {
            const int idx = tid + i * NT, r = idx >> 3, ks = idx & 7;
            int col = n0 + r;
            col = col < g.N ? col : g.N - 1;
            b16_src[i] = g.B16 + (int64_t)(g.zmod ? z % g.zmod : 0) * g.strideB16 + (int64_t)col * g.ldb16 + ks * 8;
            b16_lds[i] = BM * ROWB + r * ROWB + ((ks ^ swz(r)) << 4);
        }
    }

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        // element offsets of K row k0 in A^T / B (SRC 7 may walk a segmented K: a tile never straddles a segment, kseg % BK == 0)
        int64_t ka = (int64_t)k0 * g.lda, kb = (int64_t)k0 * g.ldb;
        if (AT && g.kseg) {
            const int seg = k0 / g.kseg, kin = k0 - seg * g.kseg;
            ka = (int64_t)seg * g.segA + (int64_t)kin * g.lda;
            kb = (int64_t)seg * g.segB + (int64_t)kin * g.ldb;
        }
        if constexpr (A16) {
#pragma unroll
            for (int i = 0; i < NA16; ++i) ra16[i] = *reinterpret_cast<const u32x4*>(a16_src[i] + k0);
        }
        if constexpr (B16) {
#pragma unroll
            for (int i = 0; i < NB16; ++i) rb16[i] = *reinterpret_cast<const u32x4*>(b16_src[i] + k0);
        }
        if constexpr (AT) {
#pragma unroll
            for (int i = 0; i < NAT; ++i)
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) rat[i][kk] = *reinterpret_cast<const f32x4*>(A + at_off[i] + ka + (int64_t)kk * g.lda);
        }
#pragma unroll
        for (int i = 0; i < ((A16 || AT) ? 0 : NA); ++i) {
            if constexpr (FAST) {
                ra[i] = *reinterpret_cast<const f32x4*>(A + a_off[i] + k0);
            } else {
                const int idx = tid + i * NT;
                const int row = m0 + (idx >> 4), k = k0 + (idx & 15) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[i][e] = (row < g.M && k + e < g.K) ? A[(int64_t)row * g.lda + k + e] : 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < (B16 ? 0 : NB); ++i) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                if constexpr (FAST) {
                    rb[i][kk] = *reinterpret_cast<const bvec*>(Bm + b_off[i] + kb + (int64_t)kk * g.ldb);
                } else {
                    const int idx = tid + i * NT;
                    const int k = k0 + (idx / NQ) * 8 + kk, col = n0 + PN * (idx % NQ);
#pragma unroll
                    for (int j = 0; j < PN; ++j)
                        rb[i][kk][j] = (k < g.K && col + j < g.N) ? Bm[(int64_t)k * g.ldb + col + j] : 0.0f;
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* S = smem16 + buf * STAGE;
        if constexpr (A16) {
#pragma unroll
            for (int i = 0; i < NA16; ++i) *reinterpret_cast<u32x4*>(S + a16_lds[i]) = ra16[i];
        }
        if constexpr (B16) {
#pragma unroll
            for (int i = 0; i < NB16; ++i) *reinterpret_cast<u32x4*>(S + b16_lds[i]) = rb16[i];
        }
        if constexpr (AT) {
#pragma unroll
            for (int i = 0; i < NAT; ++i) {
                const int q = (tid + i * NT) % NQA, ks = (tid + i * NT) / NQA;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u32x4 p;
                    p[0] = pack_bf16(rat[i][0][j], rat[i][1][j]);
                    p[1] = pack_bf16(rat[i][2][j], rat[i][3][j]);
                    p[2] = pack_bf16(rat[i][4][j], rat[i][5][j]);
                    p[3] = pack_bf16(rat[i][6][j], rat[i][7][j]);
                    const int r = 4 * q + j;
                    *reinterpret_cast<u32x4*>(S + r * ROWB + ((ks ^ swz(r)) << 4)) = p;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < ((A16 || AT) ? 0 : NA); ++i) {
            u32x2 p;
            p[0] = pack_bf16(ra[i][0], ra[i][1]);
            p[1] = pack_bf16(ra[i][2], ra[i][3]);
            *reinterpret_cast<u32x2*>(S + a_lds[i]) = p;
        }
        if constexpr (AT) {
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int j = 0; j < PN; ++j) {
                        float t4 = (rb[i][0][j] + rb[i][1][j]) + (rb[i][2][j] + rb[i][3][j]);
                        t4 += (rb[i][4][j] + rb[i][5][j]) + (rb[i][6][j] + rb[i][7][j]);
                        cs[i][j] += t4;
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < (B16 ? 0 : NB); ++i) {
            const int q = (tid + i * NT) % NQ, ks = (tid + i * NT) / NQ;
#pragma unroll
            for (int j = 0; j < PN; ++j) {       // register transpose: column j of the patch becomes 8 consecutive k
                u32x4 p;
                p[0] = pack_bf16(rb[i][0][j], rb[i][1][j]);
                p[1] = pack_bf16(rb[i][2][j], rb[i][3][j]);
                p[2] = pack_bf16(rb[i][4][j], rb[i][5][j]);
                p[3] = pack_bf16(rb[i][6][j], rb[i][7][j]);
                const int r = PN * q + j;
                *reinterpret_cast<u32x4*>(S + BM * ROWB + r * ROWB + ((ks ^ swz(r)) << 4)) = p;
            }
        }
    };

    f32x16 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    int a_row[MT], a_swz[MT], b_row[NTL], b_swz[NTL];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int r = wm * WTM + t * 32 + li;
        a_row[t] = r * ROWB;
        a_swz[t] = swz(r);
    }
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int r = wn * WTN + t * 32 + li;
        b_row[t] = BM * ROWB + r * ROWB;
        b_swz[t] = swz(r);
    }
    // which 8 k of a 16-deep MFMA step a lane supplies is free as long as A and B agree: lane half lh takes
    // 16-byte slot 2 s + lh of both images
    auto compute = [&](int buf) {
        const unsigned char* S = smem16 + buf * STAGE;
        // Fragment reads run one k-step ahead of the MFMAs that consume them: a 16-deep bf16 MFMA is only 32 cycles, so
        // an LDS round trip (~100+ cycles) in front of each group of 4 would otherwise be the critical path.
        if constexpr (DMA && MT == 2 && NTL == 1) {
            // The default instance (8 waves of 64 x 32) with hand-counted waits.  Left to the compiler, the first MFMA pair of a
            // tile waited for the reads of steps 0 AND 1 (lgkmcnt(0) instead of 3), and the pair of step 2 for the reads of step 3
            // issued just before it: ~200 cycles of stall per wave and tile that are not data dependencies.  The reads are inline
            // asm, the waits are tied to the fragment registers ("+v") so that the MFMAs stay behind them; LDS operations complete
            // in order, so the compiler's own (empty here) lgkmcnt bookkeeping is unaffected.
            const unsigned S3 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)S;
            bf16x8 fa0[2], fa1[2], fb[2];
            auto rd = [&](int s, int slot) {
                const unsigned aa0 = S3 + (unsigned)a_row[0] + (unsigned)(((2 * s + lh) ^ a_swz[0]) << 4);
                const unsigned aa1 = S3 + (unsigned)a_row[1] + (unsigned)(((2 * s + lh) ^ a_swz[1]) << 4);
                const unsigned ab = S3 + (unsigned)b_row[0] + (unsigned)(((2 * s + lh) ^ b_swz[0]) << 4);
                asm volatile("ds_read_b128 %0, %1" : "=v"(fa0[slot]) : "v"(aa0));
                asm volatile("ds_read_b128 %0, %1" : "=v"(fa1[slot]) : "v"(aa1));
                asm volatile("ds_read_b128 %0, %1" : "=v"(fb[slot]) : "v"(ab));
            };
            auto mm = [&](int slot) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[slot], fb[slot], acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[slot], fb[slot], acc[1][0], 0, 0, 0);
            };
            static_assert(BK / 16 == 4, "four k-steps per tile");
            rd(0, 0);
            rd(1, 1);
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa0[0]), "+v"(fa1[0]), "+v"(fb[0]));
            mm(0);
            __builtin_amdgcn_sched_barrier(0);      // (the MFMAs are free to sink below the later waits otherwise)
            rd(2, 0);
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa0[1]), "+v"(fa1[1]), "+v"(fb[1]));
            mm(1);
            __builtin_amdgcn_sched_barrier(0);
            rd(3, 1);
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa0[0]), "+v"(fa1[0]), "+v"(fb[0]));
            mm(0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa0[1]), "+v"(fa1[1]), "+v"(fb[1]));
            mm(1);
            return;
        }
        bf16x8 a[2][MT], b[2][NTL];
        auto read_frags = [&](int s, int slot) {
#pragma unroll
            for (int t = 0; t < MT; ++t) a[slot][t] = *reinterpret_cast<const bf16x8*>(S + a_row[t] + (((2 * s + lh) ^ a_swz[t]) << 4));
#pragma unroll
            for (int t = 0; t < NTL; ++t) b[slot][t] = *reinterpret_cast<const bf16x8*>(S + b_row[t] + (((2 * s + lh) ^ b_swz[t]) << 4));
        };
        read_frags(0, 0);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            if (s + 1 < BK / 16) read_frags(s + 1, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s & 1][mt], b[s & 1][nt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // One tile of prefetch: measured on MI355X, a second register set (two fp32 tiles = 128 KiB per block in
    // flight) changed nothing (374 vs 371 TF on conv1) -- the loop is bound by the L2 -> CU operand bandwidth
    // (~15 TB/s across the chip for this 64-KiB-per-tile-step stream), not by latency.
    if constexpr (DMA) {
        // 1-KiB pieces: piece p of an image = rows 8p .. 8p+7 (8 lanes x 16 B per row); the lane at physical slot
        // (lane & 7) fetches logical slot (lane & 7) ^ swz(row).  Pieces are dealt round-robin to the waves.
        constexpr int NWV = WM * WN, PA = BM / 8, PB = BN / 8, PPA = PA / NWV, PPB = PB / NWV;
        static_assert(PA % NWV == 0 && PB % NWV == 0, "pieces must divide over the waves");
        // sources as a wave-uniform tile base (scalar registers, advanced by the scalar unit) plus a 32-bit per-lane offset inside
        // the tile: the DMA takes the `v_off, s[base]` form, and the K loop carries no 64-bit vector pointers or adds
        uint32_t da[PPA], db[PPB];       // BYTE offsets from the tile bases (rows clamped to the matrix: < 128 ld x 2 < 2^31)
        const uint16_t* const baseA = g.A16 + (int64_t)z * g.strideA + (int64_t)m0 * g.lda;
        const uint16_t* const baseB = g.B16 + (int64_t)(g.zmod ? z % g.zmod : 0) * g.strideB16 + (int64_t)n0 * g.ldb16;
#pragma unroll
        for (int i = 0; i < PPA; ++i) {
            const int r = (wave * PPA + i) * 8 + (lane >> 3);
            const int rc = m0 + r < g.M ? r : g.M - 1 - m0;
            da[i] = 2u * ((uint32_t)((int64_t)rc * g.lda) + (uint32_t)(((lane & 7) ^ swz(r)) << 3));
        }
#pragma unroll
        for (int i = 0; i < PPB; ++i) {
            const int r = (wave * PPB + i) * 8 + (lane >> 3);
            const int rc = n0 + r < g.N ? r : g.N - 1 - n0;
            db[i] = 2u * ((uint32_t)((int64_t)rc * g.ldb16) + (uint32_t)(((lane & 7) ^ swz(r)) << 3));
        }
        // (the uniform half goes through readfirstlane: otherwise loop strength reduction turns base + k0 + offset back into a
        //  per-lane 64-bit induction pointer and the loop's DMA falls out of the scalar-base form again)
        auto uniform_ptr = [](const uint16_t* p) {
            const uint64_t v = reinterpret_cast<uint64_t>(p);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
            return reinterpret_cast<const unsigned char*>(((uint64_t)hi << 32) | lo);
        };
        auto issue = [&](int kt, int buf) {
            unsigned char* S = smem16 + buf * STAGE;
            const int k0 = kt * BK;
            const unsigned char* const ua = uniform_ptr(baseA + k0);
            const unsigned char* const ub = uniform_ptr(baseB + k0);
#pragma unroll
            for (int i = 0; i < PPA; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ua + da[i]),
                                                 (__attribute__((address_space(3))) void*)(S + (wave * PPA + i) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < PPB; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + db[i]),
                                                 (__attribute__((address_space(3))) void*)(S + BM * ROWB + (wave * PPB + i) * 1024), 16, 0, 0);
        };
        if constexpr (NS == 2) {
            issue(0, 0);
            __syncthreads();                    // carries the vmcnt(0) that retires the DMA
            W2V2_TRC();
            for (int kt = 0; kt + 1 < nk; ++kt) {
                const int cur = kt & 1;
                if (!(abl & 1)) issue(kt + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                if (!(abl & 2)) compute(cur);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                W2V2_TRC();
            }
            compute((nk - 1) & 1);
            W2V2_TRC();
        } else {
            // Ring of NS stages, NS - 1 tiles in flight.  One MFMA block per tile step is only 512 cycles while an
            // L2 / HBM round trip under load is several thousand, so a single tile of prefetch leaves the loop
            // latency-bound.  __syncthreads() would drain every DMA (vmcnt(0)); a counted wait + raw barrier keeps the
            // younger tiles in flight:  this wave's share of tile kt has landed when at most (NS - 2) tiles' worth of
            // its own DMA instructions are outstanding, and the barrier extends that to every wave's share.
            constexpr int PER_TILE = PPA + PPB;
            static_assert(PER_TILE * (NS - 2) < 64, "vmcnt is a 6-bit counter");
#pragma unroll
            for (int t = 0; t < NS - 1; ++t) issue(t < nk ? t : nk - 1, t);      // (re-reads the last tile when K is short)
            for (int kt = 0; kt < nk; ++kt) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE * (NS - 2)) : "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                const int nxt = kt + NS - 1;
                issue(nxt < nk ? nxt : nk - 1, nxt % NS);      // into the stage whose MFMAs finished before the barrier
                __builtin_amdgcn_sched_barrier(0);
                compute(kt % NS);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt + 1 < nk; ++kt) {       // last iteration peeled: the prefetch stays unconditional
            const int cur = kt & 1;
            load_tile(kt + 1);                       // tile in flight under the MFMAs below
            __builtin_amdgcn_sched_barrier(0);
            compute(cur);
            __builtin_amdgcn_sched_barrier(0);
            store_tile(cur ^ 1);                     // (round + transpose +) write the other stage
            __syncthreads();
        }
        compute((nk - 1) & 1);
    }

    if constexpr (AT) {
        if (do_colsum) {       // fold the 8 k-groups of each column through LDS (fixed order), one value per column and batch
            __syncthreads();   // every wave is done with the operand images
            float* red = reinterpret_cast<float*>(smem16);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int q = (tid + i * NT) % NQ, ks = (tid + i * NT) / NQ;
#pragma unroll
                for (int j = 0; j < PN; ++j) red[ks * BN + PN * q + j] = cs[i][j];
            }
            __syncthreads();
            if (tid < BN && n0 + tid < g.N) {
                float t8 = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) t8 += red[ks * BN + tid];
                g.colsum[(int64_t)z * g.strideCS + n0 + tid] = t8;
            }
        }
    }
    // ---- epilogue (gemm_epilogue.h): bias -> act -> + residual -> fp32 store and / or bf16 shadow ----
    const int zi = g.zmod ? z % g.zmod : z, zo = g.zmod ? z / g.zmod : 0;
    const int64_t tile_off = (int64_t)zo * g.strideC2 + (int64_t)zi * g.strideC + (int64_t)(m0 + wm * WTM) * g.ldc + (n0 + wn * WTN);
    // the operand images are dead once every wave has issued its last MFMA: the accumulators go out through wave-private
    // LDS patches so that every store writes whole 128-byte row segments (gemm_epilogue_lds)
    if (abl & 4) {
        if (acc[0][0][0] == 12345.678f) g.C[0] = 1.f;       // keep the accumulators alive
        return;
    }
    // Epilogue forms measured in round 2 (profiles/r02_gemm_bf16_study.md): these row-major C/D blocks with dword stores
    // (2 rows x 128 B per instruction) beat both C^T accumulators with 16-byte stores straight from registers (32 rows x 32 B:
    // 532 vs 613 TF on the forward mix) and C^T through wave-private LDS patches (whole 128-byte segments, 16 B per lane: 586).
    gemm_epilogue<MT, NTL, true>(acc, g.C ? g.C + tile_off : nullptr, g.C16 ? g.C16 + tile_off : nullptr,
                                 g.residual ? g.residual + tile_off : nullptr,
                                 g.bias ? g.bias + (g.zmod ? (int64_t)zi * g.strideBias : 0) + (n0 + wn * WTN) : nullptr,
                                 (int)g.ldc, g.M - (m0 + wm * WTM), g.N - (n0 + wn * WTN), g.act, li, lh);
#ifdef W2V2_TUNING
    if (trc && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (this wave's stores have left)
        trc[trc_n++] = clock64();
        trc[31] = (unsigned long long)trc_n;
    }
#endif
}

// ---- weight-gradient form dW = X^T dY with both operands by LDS-DMA and a TRANSPOSING LDS read ---------------------------
// X (K, M) and dY (K, N) lie in memory with the contraction index k as the SLOW dimension, the opposite of what an MFMA
// fragment wants (8 consecutive k per lane).  Source 7 above transposes in registers (fp32 loads, v_cvt_pk, 16-byte LDS
// stores); measured, that loop is bound by the latency of its loads at two blocks per CU (372 TF).  Here the bf16 shadows of
// both operands go into LDS exactly as they lie in memory -- rows of 128 m | n = 256 B, 64 k rows per stage, 1-KiB DMA pieces
// of 4 rows -- and the fragments are read with ds_read_b64_tr_b16 (tools/tr_read_probe.hip pins its semantics on this chip:
// within a 16-lane group lane l supplies the address of 4 consecutive columns 4 (l % 4) .. + 3 of row l / 4, and lane c
// receives rows 0 .. 3 of column c).  A lane of a 32x32x16 MFMA needs column (lane % 32) and the 8 k rows 8 (lane / 32) .. + 7:
// two such reads.  Same tile / stage / barrier structure as the forward LDS-DMA kernel (source 5).  No epilogue extras:
// C (M, N) fp32 slabs, one per batch.
__device__ __attribute__((aligned(16))) const uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};      // DMA source of the rows past validK

template <int WM, int WN, int MINB>
__global__ __launch_bounds__(WM* WN * 64, MINB) void gemm_bf16_tr_kernel(Gemm16Args g) {
    constexpr int BM = 128, BN = 128, NWV = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NTL = WTN / 32;
    constexpr int RB = 256;                         // bytes per LDS row (128 bf16)
    constexpr int IMG = BK * RB, STAGE = 2 * IMG;   // A^T image, B image: 16 KiB each
    constexpr int PIECES = IMG / 1024, PP = PIECES / NWV;     // 1-KiB pieces (4 rows) per image, per wave
    static_assert(PIECES % NWV == 0 && MT >= 1 && NTL >= 1, "bad wave grid");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;

    // Blocks go to the XCDs round-robin in dispatch order (x fastest, then z): the remap runs over the whole (slab, tile) space, so
    // that each XCD owns one contiguous run of it whatever the tile count modulo 8 is.
    const int nwg = g.tiles_m * g.tiles_n;
    int bid, zsl;
    {
        const int total = nwg * (int)gridDim.z;
        int lin = (int)blockIdx.x + nwg * (int)blockIdx.z;
        const int q = total >> 3, r = total & 7, xcd = lin & 7, idx = lin >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        zsl = lin / nwg;
        bid = lin - zsl * nwg;
    }
    // An XCD's run of nwg / 8 consecutive tiles should be as square a patch of the (tiles_m x tiles_n) grid as possible: its tiles
    // walk the K rows in step, so a patch of r rows x c columns fetches r + c operand panels into that XCD's L2 instead of 1 + r c.
    // The fastest index is therefore the SHORTER dimension (768 x 3072: 6 x 24 tiles, runs of 18 = 6 x 3 instead of 1 x 18).  PMC
    // (profiles/r03_gemm_bf16_pmc.md): this kernel moved 638 MB per launch at 5.5 TB/s with N always fastest, L2 hit rate 0.53.
    const bool m_fast = g.tiles_m < g.tiles_n;
    const int tm = m_fast ? bid % g.tiles_m : bid / g.tiles_n, tn = m_fast ? bid / g.tiles_m : bid % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = zsl;
    const uint16_t* __restrict__ Az = g.A16 + (int64_t)z * g.strideA + m0;
    const uint16_t* __restrict__ Bz = g.B16p + (int64_t)z * g.strideB + n0;
    const int nk = g.K / BK;
    // rows of this batch that exist (a select on the source address, not a branch: control flow around the DMA issue makes the
    // compiler drain vmcnt in front of the next LDS read)
    const int64_t kleft = g.validK - (int64_t)z * g.K;
    const uint16_t* const zsrc = reinterpret_cast<const uint16_t*>(g_zero16);
    int prow[PP];

    // DMA: piece p of an image = rows 4p .. 4p+3; lane -> row 4p + lane / 16, 16-byte chunk lane % 16 (8 columns)
    const uint16_t* da[PP];
    const uint16_t* db[PP];
#pragma unroll
    for (int i = 0; i < PP; ++i) {
        // bank swizzle: the four rows of a transposing read sit 256 B = one full bank sweep apart, so the 16-byte chunk index is
        // XORed with 4 (row % 4): the physical slot `lane & 15` of row r holds logical chunk (lane & 15) ^ 4 (r & 3)
        const int row = (wave * PP + i) * 4 + (lane >> 4);
        const int chunk = (lane & 15) ^ (4 * (row & 3));
        da[i] = Az + (int64_t)row * g.lda + 8 * chunk;
        db[i] = Bz + (int64_t)row * g.ldb + 8 * chunk;
        prow[i] = row;
    }
    auto issue = [&](int kt, int buf) {
        unsigned char* S = smem16 + buf * STAGE;
        const int64_t k0 = (int64_t)kt * BK;
#pragma unroll
        for (int i = 0; i < PP; ++i) {
            const uint16_t* src = k0 + prow[i] < kleft ? da[i] + k0 * g.lda : zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(S + (wave * PP + i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PP; ++i) {
            const uint16_t* src = k0 + prow[i] < kleft ? db[i] + k0 * g.ldb : zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(S + IMG + (wave * PP + i) * 1024), 16, 0, 0);
        }
    };

    // fragment addresses: group = lane / 16 -> column half (group & 1), k half = lane / 32; l = lane % 16 -> row l / 4, chunk l % 4
    const int l16 = lane & 15, grp = (lane >> 4) & 1;
    // (every row base 16 s + 8 lh [+ 4] is a multiple of 4, so row % 4 = l16 / 4 and the swizzle is a per-lane constant)
    auto swz_col = [&](int col) { const int byte = col * 2; return (((byte >> 4) ^ (4 * (l16 >> 2))) << 4) | (byte & 15); };
    int a_off[MT], b_off[NTL];
#pragma unroll
    for (int t = 0; t < MT; ++t) a_off[t] = (8 * lh + (l16 >> 2)) * RB + swz_col(wm * WTM + t * 32 + 16 * grp + 4 * (l16 & 3));
#pragma unroll
    for (int t = 0; t < NTL; ++t) b_off[t] = IMG + (8 * lh + (l16 >> 2)) * RB + swz_col(wn * WTN + t * 32 + 16 * grp + 4 * (l16 & 3));

    // The transposing reads are issued as inline asm.  Through the builtin the compiler cannot tell them from a read of the
    // stage the DMA is filling and drains vmcnt(0) in front of the first one -- i.e. it waits for the tile it has JUST requested,
    // and the loop ran DMA and MFMAs strictly one after the other.  The price is hand-placed lgkmcnt waits (tied to the fragment
    // registers through "+v" operands so that the MFMAs cannot move above them).  The compiler's own lgkmcnt bookkeeping stays
    // safe: LDS operations complete in order, so extra operations in flight only make its waits longer, never too short.
    using v2u = __attribute__((ext_vector_type(2))) unsigned;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem16;
    auto frag = [&](unsigned stage, int off, int s) -> bf16x8 {
        // rows 16 s + 8 lh + {0..3} and + {4..7}: the two halves of the 8-deep k run this lane supplies
        v2u lo, hi;
        const unsigned a0 = stage + (unsigned)off + (unsigned)(16 * s) * RB;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(a0));      // + 4 rows of 256 bytes
        union { v2u h[2]; bf16x8 v; } u;
        u.h[0] = lo;
        u.h[1] = hi;
        return u.v;
    };
    static_assert(4 * RB == 1024, "the asm offset above is 4 LDS rows");

    f32x16 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    constexpr int NREADS = 2 * (MT + NTL);      // ds_read instructions per k-step
    auto compute = [&](int buf) {
        const unsigned S = lds0 + (unsigned)buf * STAGE;
        bf16x8 a[2][MT], b[2][NTL];
        auto read_frags = [&](int s, int slot) {
#pragma unroll
            for (int t = 0; t < MT; ++t) a[slot][t] = frag(S, a_off[t], s);
#pragma unroll
            for (int t = 0; t < NTL; ++t) b[slot][t] = frag(S, b_off[t], s);
        };
        static_assert(MT == 2 && NTL == 1, "the operand lists of the waits below are written for 2 x 1 fragments");
        read_frags(0, 0);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            // wait until only the next step's reads are outstanding, and pin this step's fragments behind the wait
            if (s + 1 < BK / 16) {
                read_frags(s + 1, (s + 1) & 1);
                asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[s & 1][0]), "+v"(a[s & 1][1]), "+v"(b[s & 1][0]) : "n"(NREADS));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[s & 1][0]), "+v"(a[s & 1][1]), "+v"(b[s & 1][0]));
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s & 1][mt], b[s & 1][nt], acc[mt][nt], 0, 0, 0);
        }
    };

    issue(0, 0);
    __syncthreads();                    // carries the vmcnt(0) that retires the DMA
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int cur = kt & 1;
        issue(kt + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    compute((nk - 1) & 1);

    const int64_t tile_off = (int64_t)z * g.strideC + (int64_t)(m0 + wm * WTM) * g.ldc + (n0 + wn * WTN);
    gemm_epilogue<MT, NTL, true>(acc, g.C + tile_off, nullptr, nullptr, nullptr, (int)g.ldc, g.M - (m0 + wm * WTM), g.N - (n0 + wn * WTN),
                                 0, li, lh);
}

int launch_tr16(Gemm16Args& g, int nbatch, hipStream_t s) {
    constexpr size_t LDS = 2 * 2 * BK * 256;
    g.tiles_m = g.M / 128;
    g.tiles_n = g.N / 128;
    // 8 waves (2 x 4, 64 x 32 each): 49.6 ms per step against 49.9 with 4 waves of 64 x 64 when this kernel was introduced
    static std::atomic<bool> attr_set{false};   // (idempotent call; atomic so concurrent host threads agree on the flag)
    if (!attr_set) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_tr_kernel<2, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_set = true;
    }
    dim3 grid(g.tiles_m * g.tiles_n, 1, nbatch);
    W2V2_LAUNCH((gemm_bf16_tr_kernel<2, 4, 2>), grid, dim3(512), LDS, s, g);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

template <int SRC, int BM, int BN, int WM, int WN, int MINB>
int launch_src16(Gemm16Args& g, int nbatch, hipStream_t s) {
    constexpr size_t LDS = (SRC == 6 ? 4 : 2) * (BM + BN) * ROWB;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    static std::atomic<bool> attr_set{false};   // (idempotent call; atomic so concurrent host threads agree on the flag)
    if (!attr_set) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<SRC, BM, BN, WM, WN, MINB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_set = true;
    }
    dim3 grid(g.tiles_m * g.tiles_n, 1, nbatch), block(WM * WN * 64);
    W2V2_LAUNCH((gemm_bf16_kernel<SRC, BM, BN, WM, WN, MINB>), grid, block, LDS, s, g);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

template <int BM, int BN, int WM, int WN, int MINB>
int launch_cfg16(Gemm16Args& g, int src, int nbatch, hipStream_t s) {
    switch (src) {
        case 1: return launch_src16<1, BM, BN, WM, WN, MINB>(g, nbatch, s);
        case 2: return launch_src16<2, BM, BN, WM, WN, MINB>(g, nbatch, s);
        case 3: return launch_src16<3, BM, BN, WM, WN, MINB>(g, nbatch, s);
        case 4: return launch_src16<4, BM, BN, WM, WN, MINB>(g, nbatch, s);
        case 5: return launch_src16<5, BM, BN, WM, WN, MINB>(g, nbatch, s);
        case 6: return launch_src16<6, BM, BN, WM, WN, 1>(g, nbatch, s);
        case 7: return launch_src16<7, BM, BN, WM, WN, MINB>(g, nbatch, s);
        default: return launch_src16<0, BM, BN, WM, WN, MINB>(g, nbatch, s);
    }
}

#ifdef W2V2_TUNING
}  // namespace
unsigned long long* g_tune_trace = nullptr;      // set by w2v2_tune_set_trace (tuning_entry.hip)
namespace {
#endif

// Which shadow-fed shapes take the 128 x 256 software-pipelined kernel (measured at B = 32, profiles/r03_gemm_bf16_study.md: it wins
// 8-25 % on every model shape with N K >= 768 x 1024 and on the batched conv layers, and loses 5-10 % on the two smaller Dense
// shapes): whole 256-column tiles, at least 256 of them (half of the chip's 512 block slots), a weight of >= 576 Ki elements (round 3: 768 Ki) or a batch.  GemmShadows::force_kernel overrides
// (1 = never, 2 = whenever the operands allow it: the op-level parity tests compare the two kernels bit for bit).
bool use_sw_kernel(const GemmShadows& x, int M, int N, int K, int64_t lda, int64_t ldb16, int64_t strideA, int nbatch) {
    if (x.zmod != 0 || x.force_kernel == 1 || !gemm_bf16_sw_ok(M, N, K, lda, ldb16, strideA)) return false;
    if (x.force_kernel == 2) return true;
    if (tune_int("W2V2_GEMM16_SW", 1) == 0) return false;
    const int64_t tiles = (int64_t)((M + 127) / 128) * (N / 256) * nbatch;
    // (round 5: the threshold moved from N K >= 768 x 1024 to 768 x 768 -- with the leaner fp32 epilogue of this round the out-projection and
    //  its data gradient, 25 launches per step on the 128 x 128 kernel at MFMA busy 0.23, are faster here too: fine-tune step 32.95 -> 32.32 ms,
    //  forward 11.30 -> 11.24 ms same-box, profiles/r05_ab_sw_768x768.txt; W2V2_SW_MIN_K in the tools-only build)
    return tiles >= 256 && ((int64_t)N * K >= (int64_t)768 * tune_int("W2V2_SW_MIN_K", 768) || nbatch > 1);      // 
}

int forced_cfg16() {
    return tune_int("W2V2_GEMM16_CFG", -1);
}

}  // namespace

int launch_gemm_bf16(Profiler* prof, const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb,
                     int64_t strideB, float* C, int64_t ldc, int64_t strideC, const float* bias,
                     const float* residual, int M, int N, int K, int nbatch, int act, hipStream_t s) {
    return launch_gemm_bf16_x(prof, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, bias, residual, M, N, K, nbatch, act,
                              GemmShadows{}, s);
}

int launch_gemm_bf16_x(Profiler* prof, const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb,
                       int64_t strideB, float* C, int64_t ldc, int64_t strideC, const float* bias,
                       const float* residual, int M, int N, int K, int nbatch, int act, const GemmShadows& x,
                       hipStream_t s) {
    W2V2_REQUIRE((A || x.A16) && (B || x.B16 || x.B16p) && (C || x.C16), "gemm_bf16: null operand");
    W2V2_REQUIRE(M > 0 && N > 0 && K > 0 && nbatch > 0, "gemm_bf16: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, nbatch);
    W2V2_REQUIRE(lda >= 1 && ldb >= N && ldc >= N && ldc < (1 << 23), "gemm_bf16: bad leading dimensions");
    W2V2_REQUIRE(act >= 0 && act <= 2, "gemm_bf16: bad activation %d", act);
    Gemm16Args g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.residual = residual;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.strideA = strideA; g.strideB = strideB; g.strideC = strideC;
    g.M = M; g.N = N; g.K = K; g.act = act;
    g.A16 = x.A16; g.B16 = x.B16; g.C16 = x.C16; g.ldb16 = x.ldb16 ? x.ldb16 : K;
    g.zmod = x.zmod; g.strideB16 = x.strideB16; g.strideC2 = x.strideC2; g.strideBias = x.strideBias; g.strideB2 = x.strideB2;
    g.colsum = x.transA ? x.colsum : nullptr; g.strideCS = x.strideCS;
    g.B16p = x.B16p;
    g.kseg = x.kseg; g.segA = x.segA; g.segB = x.segB;
    W2V2_REQUIRE(x.kseg == 0 || (x.transA && !x.A16 && !x.B16p && A && B && x.kseg % BK == 0 && K % x.kseg == 0 && !x.colsum),
                 "gemm_bf16: a segmented K (kseg) is for the fp32 transposed-A form, whole segments of a multiple of 64 rows");
    g.validK = x.validK > 0 ? x.validK : (int64_t)K * nbatch;
    W2V2_REQUIRE(x.validK == 0 || (x.transA && x.A16 && x.B16p && x.validK > (int64_t)K * (nbatch - 1) + 64 * x.kextra &&
                                   x.validK <= (int64_t)K * nbatch + 64 * x.kextra),
                 "gemm_bf16: validK is for the transposed-A shadow form, and every batch must own at least one existing row");
#ifdef W2V2_TUNING
    g.abl = tune_int("W2V2_GEMM16_ABL", 0);
    g.trace = g_tune_trace;
#endif
    W2V2_REQUIRE(x.zmod >= 0 && (x.zmod == 0 || nbatch % x.zmod == 0), "gemm_bf16: batch %d is not a multiple of the inner batch %d", nbatch, x.zmod);
    const bool kfast = K % BK == 0;
    const bool a32 = A && (lda % 4 == 0) && (strideA % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool b32 = B && (N % 4 == 0) && (ldb % 4 == 0) && (strideB % 4 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    const bool a16 = x.A16 && kfast && (lda % 8 == 0) && (strideA % 8 == 0) && ((reinterpret_cast<uintptr_t>(x.A16) & 15) == 0);
    const bool b16 = x.B16 && kfast && strideB == 0 && (g.ldb16 % 8 == 0) && ((reinterpret_cast<uintptr_t>(x.B16) & 15) == 0);
    int src;
    // tuning knob: 0 = register-staged shadows (595 TF on the forward mix), 1 = LDS-DMA, 2 stages, 2 blocks/CU (617, default),
    // 2 = LDS-DMA 4-stage ring with three tiles in flight, 1 block/CU (513: deeper prefetch does not pay for half the waves)
    const int dma = tune_int("W2V2_GEMM16_DMA", 1);
    if (x.transA) {
        // both bf16 shadows, whole 128 x 128 tiles, 16-byte aligned rows: LDS-DMA + transposing LDS reads (the fp32 operands are
        // not touched and may be null)
        const int tr = tune_int("W2V2_GEMM16_TR", 1);
        if (tr && kfast && C && x.A16 && x.B16p && !x.colsum && !x.overlapA && M % 128 == 0 && N % 128 == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
            strideA % 8 == 0 && strideB % 8 == 0 && ((reinterpret_cast<uintptr_t>(x.A16) | reinterpret_cast<uintptr_t>(x.B16p)) & 15) == 0) {
            const double krows = (double)K * nbatch + 64.0 * x.kextra;          // rows of the K dimension over all slabs
            ProfScope ps(prof, FAM_GEMM_BF16, 2.0 * M * (double)N * krows, 2.0 * krows * ((double)M + N) + nbatch * 4.0 * (double)M * N, s);
            // every row exists and the columns fill 256-wide tiles: the 128 x 256 software-pipelined kernel in its transposed form
            // (gemm_bf16_sw.hip); identical bits.  Ragged row counts (validK) stay on the kernel below, which reads rows past the end as zero.
            const int64_t krows_all = (int64_t)K * nbatch + 64 * x.kextra;
            const bool whole = x.validK == 0 || x.validK == krows_all;
            // (rows missing only from the last K tile, and a zero row promised behind B: the same kernel, ragged form)
            const int krag = (!whole && x.b_zero_row && x.validK > krows_all - 64 && x.validK < krows_all) ? (int)(x.validK - (krows_all - 64)) : 0;
            if (x.force_kernel != 1 && (whole || krag > 0) && gemm_bf16_swtr_ok(M, N, K, lda, ldb, strideA, strideB) &&
                (x.force_kernel == 2 || tune_int("W2V2_GEMM16_SW", 1) != 0))
                return launch_gemm_bf16_swtr(x.A16, lda, strideA, x.B16p, ldb, strideB, C, ldc, strideC, M, N, K, nbatch, s, x.kextra, krag);
            W2V2_REQUIRE(x.kextra == 0, "gemm_bf16: uneven slabs (kextra) are a feature of the 128 x 256 transposed kernel");
            return launch_tr16(g, nbatch, s);
        }
        W2V2_REQUIRE(x.validK == 0, "gemm_bf16: validK needs the LDS-DMA weight-gradient form (both shadows, whole 128 x 128 tiles)");
        // (the generic transposed-A path below contracts exactly K * nbatch rows: an operand set that asks for more would be cut short silently)
        W2V2_REQUIRE(x.kextra == 0 && !x.b_zero_row, "gemm_bf16: uneven slabs (kextra) / the zero row behind B need the LDS-DMA weight-gradient form");
        W2V2_REQUIRE(A && kfast && b32 && (M % 4 == 0) && (lda % 4 == 0) && (strideA % 4 == 0) &&
                         ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (lda >= M || x.overlapA),
                     "gemm_bf16: transposed A needs fp32 A and B, K %% 64 == 0, M %% 4 == 0 and 16-byte alignment");
        src = 7;
    } else if (a16 && b16) src = dma == 2 ? 6 : (dma ? 5 : 4);
    else if (a16 && b32) src = 2;
    else if (b16 && a32) src = 3;
    else if (kfast && a32 && b32) src = 1;
    else src = 0;
    W2V2_REQUIRE(src != 0 || (A && B), "gemm_bf16: a shadow-only operand needs K %% 64 == 0 and 16-byte alignment");
    const double abytes = (src == 2 || (src >= 4 && src <= 6)) ? 2.0 : 4.0, bbytes = (src >= 3 && src <= 6) ? 2.0 : 4.0;
    ProfScope ps(prof, FAM_GEMM_BF16, 2.0 * M * (double)N * K * nbatch,
                 nbatch * (abytes * (double)M * K + (C ? 4.0 : 0.0) * (double)M * N + (x.C16 ? 2.0 : 0.0) * (double)M * N) +
                     bbytes * (double)K * N, s);
    int cfg = forced_cfg16();
    // both operands from shadows, enough tiles to fill the chip: 128 x 256 tiles, 4-wave software-pipelined blocks, two per CU
    // (gemm_bf16_sw.hip).  Same bits as the kernels below.
    if (src == 5 && use_sw_kernel(x, M, N, K, lda, g.ldb16, strideA, nbatch))
        return launch_gemm_bf16_sw(x.A16, lda, strideA, x.B16, g.ldb16, C, x.C16, ldc, strideC, bias, residual, M, N, K, nbatch, act, s);
    if (cfg == 2 && src == 1) return launch_src16<1, 256, 256, 2, 4, 1>(g, nbatch, s);   // tile study: 8 waves of 128x64
    if (N <= 64 && src == 5) return launch_src16<5, 128, 64, 2, 2, 2>(g, nbatch, s);     // narrow outputs (grouped conv: 48 | 64 columns)
    if (N <= 64 && src == 7) return launch_src16<7, 128, 64, 2, 2, 2>(g, nbatch, s);
    // both operands from shadows by LDS-DMA: 8 waves of 64x32 per 128x128 tile (2 x 4 per SIMD pair, 2 blocks / CU) measured
    // 670 TF on the forward mix against 642 for 4 waves of 64x64 (more waves issuing DMA pieces and fragment reads); the
    // register-staged sources keep 4 waves (their patch ownership is tied to 256 threads)
    if (src == 5 && cfg != 0) return launch_src16<5, 128, 128, 2, 4, 2>(g, nbatch, s);
    return launch_cfg16<128, 128, 2, 2, 2>(g, src, nbatch, s);
}

}  // namespace w2v2
