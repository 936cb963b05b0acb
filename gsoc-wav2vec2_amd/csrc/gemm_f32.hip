// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
//
// One kernel template serves every dense contraction of the path:
//   * Dense layers (q|k|v, out, FFN, projection, lm_head)      -- encoder.py:15-18,99-104
//   * strided Conv1D layers 1..6 as an implicit GEMM           -- feature_extractor.py:31-37
// In channels-last layout the conv window of output frame t is the CONTIGUOUS run
// in[t*stride*C .. t*stride*C + K*C), so a strided conv is exactly a GEMM whose A
// matrix has leading dimension lda = stride*C < K*C (overlapping rows) and whose
// B matrix is the (K*C_in, C_out) reshape of the TF kernel.  No im2col copy exists.
//
// Tiling: BM x BN x 32 block tile, WM x WN waves, each wave a (BM/WM) x (BN/WN)
// sub-tile of 32x32 MFMA accumulators.  fp32 MFMA issues at 64 cycles per SIMD, so
// LDS traffic is far from the limit; what matters is (a) keeping the matrix pipe
// issuing back-to-back (register-prefetched global loads, double-buffered LDS, one
// barrier per K tile) and (b) the L2->CU operand traffic, which scales with
// 1/BM + 1/BN: the 128x128 tile asks ~4.9 TB/s of L2 at the fp32 peak, the
// 128x256 / 256x128 tiles ~3.7 TB/s.
//
// The k-pairing inside an MFMA (which two k-indices one 32x32x2 step consumes)
// is free as long as A and B agree, so each lane fetches its A fragment as ONE
// 16-byte LDS read (4 consecutive k) and the k-steps are taken as
// {k, k+4}, {k+1, k+5}, ... within an 8-wide k block.
#include <stdlib.h>


#include "common.h"
#include "gemm_epilogue.h"

namespace w2v2 {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;   // native vector: arrays of HIP's float4 struct land in scratch

namespace {

constexpr int BK = 32;
constexpr int LDA_S = BK + 4;   // +16 B row pad: conflict-free ds_read_b128 of a column slice

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* residual;
    int64_t lda, ldb, ldc, strideA, strideB, strideC;
    int M, N, K, act;
    int tiles_m, tiles_n;
    int gm = 0;      // grouped tile order of the LDS-DMA kernel: rows per group (common.h::grouped_tile); 0 = linear order
};

__device__ __forceinline__ float ld_a(const float* A, const GemmArgs& g, int row, int k) {
    return (row < g.M && k < g.K) ? A[(int64_t)row * g.lda + k] : 0.0f;
}
__device__ __forceinline__ float ld_b(const float* B, const GemmArgs& g, int k, int col) {
    return (k < g.K && col < g.N) ? B[(int64_t)k * g.ldb + col] : 0.0f;
}


template <int BM, int BN, int WM, int WN>
struct Cfg {
    static constexpr int NT = WM * WN * 64;             // threads
    static constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile
    static constexpr int MT = WTM / 32, NTL = WTN / 32; // 32x32 accumulators per wave
    static constexpr int NA = BM * (BK / 4) / NT;       // float4 per thread per A tile
    static constexpr int NB = BK * (BN / 4) / NT;       // float4 per thread per B tile
    static constexpr int STAGE = BM * LDA_S + BK * BN;  // floats per LDS stage
    static constexpr size_t LDS = 2 * STAGE * sizeof(float);
};

template <bool FAST, int BM, int BN, int WM, int WN, int MINW>
__global__ __launch_bounds__(WM* WN * 64, MINW) void gemm_f32_kernel(GemmArgs g) {
    using C_ = Cfg<BM, BN, WM, WN>;
    constexpr int NT = C_::NT, MT = C_::MT, NTL = C_::NTL, NA = C_::NA, NB = C_::NB, STAGE = C_::STAGE;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware tile order: the dispatcher places block b on XCD b % 8; give each
    // XCD a contiguous run of tiles (N fastest) so a BM-row A panel is re-read
    // from that XCD's L2 by the column tiles next to it.  Bijective for any count.
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
    const float* __restrict__ Bm = g.B + (int64_t)z * g.strideB;

    // ---- global -> register staging of one K tile (float4 per thread) ---------
    f32x4 ra[NA], rb[NB];
    int64_t a_off[NA], b_off[NB];
    int a_lds[NA], b_lds[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int idx = tid + i * NT;
        const int r = idx / (BK / 4), c = (idx % (BK / 4)) * 4;
        int row = m0 + r;
        row = row < g.M ? row : g.M - 1;               // clamp: loads stay in bounds, stores are guarded
        a_off[i] = (int64_t)row * g.lda + c;
        a_lds[i] = r * LDA_S + c;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idx = tid + i * NT;
        const int r = idx / (BN / 4), c = (idx % (BN / 4)) * 4;
        int col = n0 + c;
        col = col < g.N ? col : (g.N >= 4 ? g.N - 4 : 0);   // clamped columns feed accumulators never stored
        b_off[i] = (int64_t)r * g.ldb + col;
        b_lds[i] = BM * LDA_S + r * BN + c;
    }

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if constexpr (FAST) {
                ra[i] = *reinterpret_cast<const f32x4*>(A + a_off[i] + k0);
            } else {
                const int idx = tid + i * NT;
                const int row = m0 + idx / (BK / 4), k = k0 + (idx % (BK / 4)) * 4;
                ra[i] = f32x4{ld_a(A, g, row, k), ld_a(A, g, row, k + 1), ld_a(A, g, row, k + 2), ld_a(A, g, row, k + 3)};
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if constexpr (FAST) {
                rb[i] = *reinterpret_cast<const f32x4*>(Bm + b_off[i] + (int64_t)k0 * g.ldb);
            } else {
                const int idx = tid + i * NT;
                const int k = k0 + idx / (BN / 4), col = n0 + (idx % (BN / 4)) * 4;
                rb[i] = f32x4{ld_b(Bm, g, k, col), ld_b(Bm, g, k, col + 1), ld_b(Bm, g, k, col + 2), ld_b(Bm, g, k, col + 3)};
            }
        }
    };
    auto store_tile = [&](int buf) {
        float* S = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(S + a_lds[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(S + b_lds[i]) = rb[i];
    };

    f32x16 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Fragment reads are left to hipcc's own just-in-time placement (ds_read2_b32 + counted lgkmcnt in
    // front of each group of 4 MFMAs): with two waves per SIMD the partner wave covers those waits.
    // An explicitly software-pipelined variant (reads one k block ahead, pinned with sched_barrier)
    // measured 1 % SLOWER at 128x128 and 25 % slower at 256x256 (VGPR 246).
    auto compute = [&](int buf) {
        const float* As = smem + buf * STAGE + (wm * C_::WTM + li) * LDA_S + 4 * lh;
        const float* Bs = smem + buf * STAGE + BM * LDA_S + (4 * lh) * BN + wn * C_::WTN + li;
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            f32x4 a[MT];
            float b[NTL][4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(As + mt * 32 * LDA_S + kb * 8);
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) b[nt][e] = Bs[(kb * 8 + e) * BN + nt * 32];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTL; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][e], b[nt][e], acc[mt][nt], 0, 0, 0);
        }
    };

    const int nk = (g.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    // steady state: the prefetch is unconditional (last iteration peeled) so the staging
    // registers are plain SSA values -- a conditional prefetch makes hipcc spill them to scratch
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int cur = kt & 1;
        load_tile(kt + 1);              // in flight under the MFMAs below
        // hipcc otherwise sinks the prefetch below the MFMA block (to save 32 VGPRs) and then waits
        // for it at once: the whole L2/HBM latency exposed every K tile.  Pin the order.
        __builtin_amdgcn_sched_barrier(0);
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);
        store_tile(cur ^ 1);
        __syncthreads();
    }
    compute((nk - 1) & 1);

    // ---- epilogue: bias -> activation -> + residual -> store (gemm_epilogue.h) ----
    {
        const int64_t tile_off = (int64_t)z * g.strideC + (int64_t)(m0 + wm * C_::WTM) * g.ldc + (n0 + wn * C_::WTN);
        gemm_epilogue<MT, NTL, false>(acc, g.C + tile_off, nullptr, g.residual ? g.residual + tile_off : nullptr,
                                     g.bias ? g.bias + (n0 + wn * C_::WTN) : nullptr, (int)g.ldc, g.M - (m0 + wm * C_::WTM),
                                     g.N - (n0 + wn * C_::WTN), g.act, li, lh);
    }
}

// ---- LDS-DMA variant (FAST shapes only) ---------------------------------------
// Same 128x128x32 tile and MFMA loop, but the K tiles go HBM/L2 -> LDS directly with
// global_load_lds_dwordx4 (no VGPR staging, no ds_write pass, 32 fewer VGPRs).  The DMA writes
// LDS lane-linearly (wave-uniform base + lane * 16 B), so the A image cannot be row-padded; the
// bank-conflict fix is an XOR swizzle applied on the per-lane GLOBAL source address and again on
// the fragment read:  k-slot' = k-slot ^ ((row >> 1) & 7)   (16-B slots, 8 per 128-B row).
typedef __attribute__((address_space(3))) float lds_f32;
typedef const __attribute__((address_space(1))) float glb_f32;

__device__ __forceinline__ void dma16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((glb_f32*)g, (lds_f32*)l, 16, 0, 0);
}

template <int WM, int WN, int MINW, int BKT, int BM = 128, int BN = 128>
__global__ __launch_bounds__(WM* WN * 64, MINW) void gemm_f32_dma_kernel(GemmArgs g) {
    constexpr int STAGE = BM * BKT + BKT * BN;   // 32 KiB at 128x128x32, 64 KiB at 256x256x32
    constexpr int RPP = 256 / BKT;              // A rows per 1-KiB DMA piece (8 | 16)
    constexpr int SPR = BKT / 4;                // 16-B k-slots per A row (8 | 4)
    constexpr int SW = BKT == 32 ? 1 : 2;       // swizzle: slot ^= (row >> SW) & (SPR - 1)
    constexpr int NPA = BM * BKT / 256;         // 1-KiB pieces of the A tile
    constexpr int NPB = BKT * BN / 256;         // 1-KiB pieces of the B tile
    constexpr int NW = WM * WN;                 // waves per block
    constexpr int PPA = (NPA + NW - 1) / NW, PPB = (NPB + NW - 1) / NW;   // pieces per wave per K tile (round-robin; the
                                                                          // last may be absent when NW does not divide)
    constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NTL = WTN / 32;
    static_assert(PPA >= 1 && PPB >= 1 && MT >= 1 && NTL >= 1, "bad wave grid");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    const int nwg = g.tiles_m * g.tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware order: each XCD owns a contiguous run of tiles, N fastest (three other orders measured within 1 %)
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm = bid / g.tiles_n, tn = bid % g.tiles_n;
    if (g.gm > 0) grouped_tile(bid, g.tiles_m, g.tiles_n, g.gm, tm, tn);      // wide outputs: gm x (inflight / gm) patches per XCD (common.h)
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.z;
    const float* __restrict__ A = g.A + (int64_t)z * g.strideA;
    const float* __restrict__ Bm = g.B + (int64_t)z * g.strideB;

    // per-lane global sources of this wave's A pieces and B pieces (1 KiB each)
    const float* a_src[PPA];
    const float* b_src[PPB];
#pragma unroll
    for (int i = 0; i < PPA; ++i) {
        const int piece = min(wave + i * NW, NPA - 1);      // pieces of RPP rows
        const int r = piece * RPP + lane / SPR;             // A row inside the tile
        int row = m0 + r;
        row = row < g.M ? row : g.M - 1;
        const int slot = (lane % SPR) ^ ((r >> SW) & (SPR - 1));   // swizzled 16-B k-slot this lane fetches
        a_src[i] = A + (int64_t)row * g.lda + slot * 4;
    }
#pragma unroll
    for (int i = 0; i < PPB; ++i) {
        const int flat = min(wave + i * NW, NPB - 1) * 256 + lane * 4;  // lane-linear position inside the (BKT, BN) tile
        const int br = flat / BN;
        int col = n0 + flat % BN;
        col = col < g.N ? col : (g.N >= 4 ? g.N - 4 : 0);
        b_src[i] = Bm + (int64_t)br * g.ldb + col;
    }
    auto issue_tile = [&](int kt, int buf) {
        float* S = smem + buf * STAGE;
        const int k0 = kt * BKT;
#pragma unroll
        for (int i = 0; i < PPA; ++i)
            if (NPA % NW == 0 || wave + i * NW < NPA) dma16(a_src[i] + k0, S + (wave + i * NW) * 256);
#pragma unroll
        for (int i = 0; i < PPB; ++i)
            if (NPB % NW == 0 || wave + i * NW < NPB) dma16(b_src[i] + (int64_t)k0 * g.ldb, S + BM * BKT + (wave + i * NW) * 256);
    };

    f32x16 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment read offsets: row i = wm*WTM + mt*32 + li, k-slot j = 2 kb + lh, slot' = j ^ ((i >> 1) & 7)
    int a_row[MT], a_swz[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int i = wm * WTM + mt * 32 + li;
        a_row[mt] = i * BKT;
        a_swz[mt] = (i >> SW) & (SPR - 1);
    }
    auto compute = [&](int buf) {
        const float* As = smem + buf * STAGE;
        const float* Bs = smem + buf * STAGE + BM * BKT + (4 * lh) * BN + wn * WTN + li;
#pragma unroll
        for (int kb = 0; kb < BKT / 8; ++kb) {
            f32x4 a[MT];
            float b[NTL][4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                a[mt] = *reinterpret_cast<const f32x4*>(As + a_row[mt] + (((2 * kb + lh) ^ a_swz[mt]) << 2));
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) b[nt][e] = Bs[(kb * 8 + e) * BN + nt * 32];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTL; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][e], b[nt][e], acc[mt][nt], 0, 0, 0);
        }
    };

    const int nk = g.K / BKT;
    issue_tile(0, 0);
    __syncthreads();                    // carries the vmcnt(0) that retires the DMA
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int cur = kt & 1;
        issue_tile(kt + 1, cur ^ 1);    // DMA into the other buffer, in flight under the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    compute((nk - 1) & 1);

    {
        const int64_t tile_off = (int64_t)z * g.strideC + (int64_t)(m0 + wm * WTM) * g.ldc + (n0 + wn * WTN);
        gemm_epilogue<MT, NTL, false>(acc, g.C + tile_off, nullptr, g.residual ? g.residual + tile_off : nullptr,
                                     g.bias ? g.bias + (n0 + wn * WTN) : nullptr, (int)g.ldc, g.M - (m0 + wm * WTM),
                                     g.N - (n0 + wn * WTN), g.act, li, lh);
    }
}

template <int WM, int WN, int MINW, int BKT = 32, int BM = 128, int BN = 128>
int launch_dma(GemmArgs& g, int nbatch, hipStream_t s) {
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    dim3 grid(g.tiles_m * g.tiles_n, 1, nbatch), block(WM * WN * 64);
    const size_t lds = 2 * (BM * BKT + BKT * BN) * sizeof(float);
    // tiles an XCD has in flight: 32 CUs x the blocks of this instance a CU holds (LDS-bound; at least MINW / waves-per-SIMD)
    const int per_cu = (int)(160 * 1024 / lds) < 1 ? 1 : (int)(160 * 1024 / lds) > 4 ? 4 : (int)(160 * 1024 / lds);
    g.gm = nbatch == 1 ? tile_group_rows(g.tiles_m, g.tiles_n, (int64_t)BM * g.K * 4, 32 * per_cu) : 0;
    static std::atomic<bool> attr_set{false};   // (idempotent call; atomic so concurrent host threads agree on the flag)
    if (!attr_set && lds > 64 * 1024) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_dma_kernel<WM, WN, MINW, BKT, BM, BN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    W2V2_LAUNCH((gemm_f32_dma_kernel<WM, WN, MINW, BKT, BM, BN>), grid, block, lds, s, g);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

template <int BM, int BN, int WM, int WN, int MINW>
int launch_cfg(GemmArgs& g, bool fast, int nbatch, hipStream_t s) {
    using C_ = Cfg<BM, BN, WM, WN>;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    static std::atomic<bool> attr_set{false};   // (idempotent call; atomic so concurrent host threads agree on the flag)
    if (!attr_set) {   // > 64 KiB of dynamic LDS needs the opt-in
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<true, BM, BN, WM, WN, MINW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)C_::LDS));
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<false, BM, BN, WM, WN, MINW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)C_::LDS));
        attr_set = true;
    }
    dim3 grid(g.tiles_m * g.tiles_n, 1, nbatch), block(C_::NT);
    if (fast)
        W2V2_LAUNCH((gemm_f32_kernel<true, BM, BN, WM, WN, MINW>), grid, block, C_::LDS, s, g);
    else
        W2V2_LAUNCH((gemm_f32_kernel<false, BM, BN, WM, WN, MINW>), grid, block, C_::LDS, s, g);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int forced_cfg() {
    return tune_int("W2V2_GEMM_CFG", -1);
}

// ---- split-K for small problems (serving: B = 1) ------------------------------------------------------------------
// A single utterance gives the N = 768 GEMMs 36 tiles of 64x64 for 256 CUs, each walking a K = 3072 loop of 96 steps
// alone.  Splitting K over S batches (the batched-strides contract of this kernel: A advances along its columns, B along
// its rows, C is a slab) fills the chip; this kernel then folds the slabs with the epilogue the GEMM skipped.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C, const float* __restrict__ bias,
                                                            const float* __restrict__ residual, int M, int N, int64_t ldc, int S, int act) {
    const int64_t n4 = (int64_t)M * N / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t e = i * 4, row = e / N;
        const int col = (int)(e % N);
        float4 acc = reinterpret_cast<const float4*>(slabs)[i];
        for (int z = 1; z < S; ++z) {
            const float4 v = reinterpret_cast<const float4*>(slabs + (int64_t)z * M * N)[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (bias) { acc.x += bias[col]; acc.y += bias[col + 1]; acc.z += bias[col + 2]; acc.w += bias[col + 3]; }
        acc.x = apply_act(acc.x, act); acc.y = apply_act(acc.y, act); acc.z = apply_act(acc.z, act); acc.w = apply_act(acc.w, act);
        if (residual) {
            const float4 r = *reinterpret_cast<const float4*>(residual + row * ldc + col);
            acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
        }
        *reinterpret_cast<float4*>(C + row * ldc + col) = acc;
    }
}

thread_local int tl_precision = 0;

}  // namespace

void gemm_set_precision(int mode) { tl_precision = mode; }
int gemm_get_precision() { return tl_precision; }

int launch_gemm(Profiler* prof, const float* A, int64_t lda, int64_t strideA, const float* B,
                int64_t ldb, float* C, int64_t ldc, int64_t strideC, const float* bias,
                const float* residual, int M, int N, int K, int nbatch, int act, hipStream_t s) {
    return launch_gemm_ex(prof, A, lda, strideA, B, ldb, 0, C, ldc, strideC, bias, residual, M, N, K, nbatch, act, s);
}

// strideB != 0: every batch has its own B (split-K weight gradients: A, B advance along K, C is a slab)
int launch_gemm_ex(Profiler* prof, const float* A, int64_t lda, int64_t strideA, const float* B,
                   int64_t ldb, int64_t strideB, float* C, int64_t ldc, int64_t strideC, const float* bias,
                   const float* residual, int M, int N, int K, int nbatch, int act, hipStream_t s) {
    if (tl_precision == 1)
        return launch_gemm_bf16(prof, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, bias, residual, M, N, K, nbatch, act, s);
    W2V2_REQUIRE(A && B && C, "gemm: null operand");
    W2V2_REQUIRE(M > 0 && N > 0 && K > 0 && nbatch > 0, "gemm: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, nbatch);
    W2V2_REQUIRE(lda >= 1 && ldb >= N && ldc >= N && ldc < (1 << 23), "gemm: bad leading dimensions");
    W2V2_REQUIRE(act >= 0 && act <= 2, "gemm: bad activation %d", act);
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.residual = residual;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.strideA = strideA; g.strideB = strideB; g.strideC = strideC;
    g.M = M; g.N = N; g.K = K; g.act = act;
    const bool fast = (K % BK == 0) && (N % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                      (strideA % 4 == 0) && (strideB % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    ProfScope ps(prof, FAM_GEMM, 2.0 * M * (double)N * K * nbatch,
                 4.0 * nbatch * ((double)M * K + (double)M * N) + 4.0 * (double)K * N, s);
    int cfg = forced_cfg();
    // default: the LDS-DMA 128x128 kernel with 8 waves (2x4, each wave 64x32), 2 blocks per CU = 4 waves
    // per SIMD, whenever the shape allows 16-byte global accesses (every GEMM of the model does).
    // Measured on MI355X, B=32 base shapes (profiles/r01_gemm_tile_study.md): this 118-129 TF; the same
    // with 4 waves 112-123; register-staged 128x128 107-120; 128x256 / 256x128 at 1 wave per SIMD 75-100;
    // 256x256 8-wave 67-115; BK=16 variants with 3-4 blocks per CU 114-125.
    // (cfg 14 = 128x96 tiles with 6 waves, meant to turn the 2.25 "rounds" of the N = 768 GEMMs into 3.0 exact ones:
    // measured 100 TF against 114-121 -- blocks are not scheduled in lock-step rounds, so the quantisation it removes
    // does not exist, and the 6-wave block is simply less efficient.)
    if (cfg < 0 && fast && nbatch == 1 && strideB == 0 && K >= 1024 && (ldc % 4) == 0 &&
        ((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0) {
        const int64_t tiles64 = (int64_t)((M + 63) / 64) * ((N + 63) / 64);
        int S = 1;
        for (int cand = 8; cand >= 2; cand >>= 1)
            if (K % (cand * BK) == 0 && K / cand >= 256 && tiles64 * cand <= 512) { S = cand; break; }
        if (S > 1 && tiles64 <= 256) {
            const size_t need = (size_t)S * M * N;
            float* ws = nullptr;
            void* raw = nullptr;
            if (int e = stream_scratch(SCRATCH_SPLITK, s, need * sizeof(float), &raw)) return e;
            ws = reinterpret_cast<float*>(raw);
            GemmArgs h = g;
            h.C = ws; h.bias = nullptr; h.residual = nullptr; h.act = 0;
            h.K = K / S; h.strideA = K / S; h.strideB = (int64_t)(K / S) * ldb; h.ldc = N; h.strideC = (int64_t)M * N;
            if (int e = launch_dma<2, 2, 2, 32, 64, 64>(h, S, s)) return e;
            const int64_t n4 = (int64_t)M * N / 4;
            int64_t blocks = (n4 + 255) / 256;
            blocks = blocks > 2048 ? 2048 : blocks;
            W2V2_LAUNCH(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ws, C, bias, residual, M, N, ldc, S, act);
            W2V2_HIP_CHECK(hipGetLastError());
            return W2V2_OK;
        }
    }
    if (cfg < 0) {
        cfg = 7;
        // small problems (batch 1-4 of the transformer GEMMs: M = 768 rows is 6 row tiles): 128x128 tiles leave most
        // of the 256 CUs idle and serialise a long K loop per tile, so switch to 64x64 tiles (4x the blocks, 4 waves)
        const int64_t tiles128 = (int64_t)((M + 127) / 128) * ((N + 127) / 128) * nbatch;
        if (fast && tiles128 < 384) cfg = 16;
        // Tail fill.  512 blocks of the 128x128 kernel are resident (2 per CU); T tiles whose last, partial round fills at
        // most a quarter of those slots (N = 768 at B = 32: 1152 = 2 x 512 + 128) leave most CUs idle for one whole tile time.
        // (A half-full last round is better left alone: large-robust at B = 16 has 768 = 512 + 256 tiles for N = 1024,
        // a third of the work would move to the slower 64x64 kernel, 82.3 vs 80.5 ms per forward.)
        // The rows of that partial round are computed with 64x64 tiles instead (4x the blocks, a quarter of the time each).
        // Every output element still sums its K products in the same order, so results do not depend on the tiling
        // (a row's value is independent of its position in the batch: test_linearity_of_lm_head_at_full_size).
        const int tail_knob = tune_int("W2V2_GEMM_TAIL", 1);
        const int64_t tn = (N + 127) / 128, S = 512, r = tiles128 % S;
        if (tail_knob && fast && cfg == 7 && nbatch == 1 && tiles128 > S && r != 0 && 4 * r <= S) {
            const int64_t main_rows = ((tiles128 - r) / tn) * 128;
            if (main_rows > 0 && main_rows < M) {
                GemmArgs h = g;
                h.M = (int)main_rows;
                if (int e = launch_dma<2, 4, 2>(h, 1, s)) return e;
                GemmArgs t = g;
                t.A = g.A + main_rows * lda;
                t.C = g.C + main_rows * ldc;
                t.residual = g.residual ? g.residual + main_rows * ldc : nullptr;
                t.M = M - (int)main_rows;
                // (Round 4 measured the alternative the round-3 review asked for -- the leftover rows of the K = 3072 members on the 128 x 128
                //  kernel, K split four ways into slabs folded by splitk_reduce_kernel: 512 blocks at full occupancy, 24 k-steps each.
                //  Forward 62.38 / 62.44 ms against 62.32 / 62.47 with the 64 x 64 tiles, arms interleaved on one box, two-way split
                //  62.8-62.9 (profiles/r04_ab_gemm_tail_splitk.txt): a quarter-length K loop pays prologue + epilogue + the fold of 25 MB
                //  of slabs, which is what the small tiles lose to their single accumulator per wave.  Not kept: it also gave up the
                //  bit-for-bit independence of a row from its position in the batch.)
                return launch_dma<2, 2, 2, 32, 64, 64>(t, 1, s);
            }
        }
    }
    // Wide tiles for the well-filled shapes (round 4, profiles/r04_gemm_f32_tile_study.txt): 256 x 128 x 16, 8 waves of 64 x 64 (4 x 2), still
    // two blocks per CU (48 KiB of LDS each, 118 VGPRs) -- 25 % less LDS-DMA per flop and a third fewer fragment reads per MFMA than
    // 128 x 128 x 32, the same flops per barrier.  Measured per shape against the default: conv1 132.6 vs 130.4 TF, conv2 131.1 vs 130.5,
    // q|k|v 129.3 vs 127.4, FFN up 126.5 vs 125.3; it LOSES where its tile count quantises (N = 768: 99-104 vs 115-121; conv3-6 with their
    // 4-8 % row padding), so it takes only GEMMs with >= 3 full rounds of tiles and <= 2.5 % padded rows.  Every output element still sums
    // its K products in ascending order two at a time: the same bits as the 128 x 128 kernel.
    const int wide_min = tune_int("W2V2_GEMM_WIDE", 1536);      // fewest 256 x 128 tiles that take the wide kernel (0: never)
    if (cfg == 7 && fast && wide_min > 0 && N % 128 == 0) {
        const int64_t tm256 = (M + 255) / 256, tiles256 = tm256 * (N / 128) * nbatch;
        if (tiles256 >= wide_min && (tm256 * 256 - M) * 40 <= M) return launch_dma<4, 2, 2, 16, 256, 128>(g, nbatch, s);
    }
    switch (cfg) {
        case 16: if (fast) return launch_dma<2, 2, 2, 32, 64, 64>(g, nbatch, s); return launch_cfg<128, 128, 2, 2, 2>(g, fast, nbatch, s);
        case 14: if (fast) return launch_dma<2, 3, 2, 32, 128, 96>(g, nbatch, s); return launch_cfg<128, 128, 2, 2, 2>(g, fast, nbatch, s);
        case 1: return launch_cfg<128, 256, 2, 2, 1>(g, fast, nbatch, s);
        case 2: return launch_cfg<256, 128, 2, 2, 1>(g, fast, nbatch, s);
        case 3: return launch_cfg<256, 256, 4, 2, 2>(g, fast, nbatch, s);
        case 4: if (fast) return launch_dma<2, 2, 2>(g, nbatch, s); return launch_cfg<128, 128, 2, 2, 2>(g, fast, nbatch, s);
        case 6: if (fast) return launch_dma<4, 2, 2>(g, nbatch, s); return launch_cfg<128, 128, 2, 2, 2>(g, fast, nbatch, s);
        case 7: if (fast) return launch_dma<2, 4, 2>(g, nbatch, s); return launch_cfg<128, 128, 2, 2, 2>(g, fast, nbatch, s);
        case 9: if (fast) return launch_dma<2, 4, 3, 16>(g, nbatch, s); return launch_cfg<128, 128, 2, 2, 2>(g, fast, nbatch, s);
        case 10: if (fast) return launch_dma<2, 4, 4, 16>(g, nbatch, s); return launch_cfg<128, 128, 2, 2, 2>(g, fast, nbatch, s);
        case 11: if (fast) return launch_dma<2, 2, 4, 16>(g, nbatch, s); return launch_cfg<128, 128, 2, 2, 2>(g, fast, nbatch, s);
        case 12: if (fast && g.N >= 256) return launch_dma<4, 4, 1, 32, 256, 256>(g, nbatch, s); return launch_cfg<128, 128, 2, 2, 2>(g, fast, nbatch, s);
        case 13: if (fast && g.N >= 128) return launch_dma<4, 2, 1, 32, 256, 128>(g, nbatch, s); return launch_cfg<128, 128, 2, 2, 2>(g, fast, nbatch, s);
        case 8: if (fast) return launch_dma<4, 4, 1>(g, nbatch, s); return launch_cfg<128, 128, 2, 2, 2>(g, fast, nbatch, s);
#ifdef W2V2_TUNING
        // round-4 tile study (tools-only build): wider tiles at BK = 16, two blocks per CU -- 25 % less LDS-DMA per flop than 128 x 128 x 32
        case 17: if (fast && g.N >= 256) return launch_dma<2, 4, 2, 16, 128, 256>(g, nbatch, s); return launch_dma<2, 4, 2>(g, nbatch, s);
        case 18: if (fast && g.N >= 128) return launch_dma<4, 2, 2, 16, 256, 128>(g, nbatch, s); return launch_dma<2, 4, 2>(g, nbatch, s);
        case 19: if (fast && g.N >= 256) return launch_dma<2, 4, 2, 32, 128, 256>(g, nbatch, s); return launch_dma<2, 4, 2>(g, nbatch, s);
#endif
        default: return launch_cfg<128, 128, 2, 2, 2>(g, fast, nbatch, s);
    }
}

}  // namespace w2v2
