// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
//
// One kernel serves every dense contraction of the path:
//   * Dense layers (q|k|v, out, FFN, projection, lm_head)      -- encoder.py:15-18,99-104
//   * strided Conv1D layers 1..6 as an implicit GEMM           -- feature_extractor.py:31-37
// In channels-last layout the conv window of output frame t is the CONTIGUOUS run
// in[t*stride*C .. t*stride*C + K*C), so a strided conv is exactly a GEMM whose A
// matrix has leading dimension lda = stride*C < K*C (overlapping rows) and whose
// B matrix is the (K*C_in, C_out) reshape of the TF kernel.  No im2col copy exists.
//
// Tiling: 128x128x32 block tile, 256 threads = 4 waves as 2x2, each wave a 64x64
// sub-tile = 2x2 MFMA 32x32 accumulators (64 acc VGPRs).  fp32 MFMA issues at
// 64 cycles per SIMD, so LDS traffic (2 b128 + 8 b32 reads per 16 MFMAs) is far
// from the limit; the design goal is simply to keep the matrix pipe issuing
// back-to-back: register-prefetched global loads, double-buffered LDS, one
// barrier per K tile, 2 blocks per CU.
//
// The k-pairing inside an MFMA (which two k-indices one 32x32x2 step consumes)
// is free as long as A and B agree, so each lane fetches its A fragment as ONE
// 16-byte LDS read (4 consecutive k) and the k-steps are taken as
// {k, k+4}, {k+1, k+5}, ... within an 8-wide k block.
#include "common.h"

namespace w2v2 {

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDA_S = BK + 4;   // +16 B row pad: conflict-free ds_read_b128 of a column slice
constexpr int LDB_S = BN;       // B fragments are row-contiguous b32 reads: no pad needed
constexpr int STAGE_FLOATS = BM * LDA_S + BK * LDB_S;

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* residual;
    int64_t lda, ldb, ldc, strideA, strideC;
    int M, N, K, act;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ float ld_a(const float* A, const GemmArgs& g, int row, int k) {
    return (row < g.M && k < g.K) ? A[(int64_t)row * g.lda + k] : 0.0f;
}
__device__ __forceinline__ float ld_b(const float* B, const GemmArgs& g, int k, int col) {
    return (k < g.K && col < g.N) ? B[(int64_t)k * g.ldb + col] : 0.0f;
}

template <bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware tile order: the dispatcher places block b on XCD b % 8; give each
    // XCD a contiguous run of tiles (N fastest) so a 128-row A panel is re-read
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
    const float* __restrict__ Bm = g.B;

    // ---- global -> register staging of one K tile ---------------------------
    // A tile 128x32: thread -> rows (tid>>3)+32i, float4 column (tid&7)
    // B tile 32x128: thread -> rows (tid>>5)+8i,  float4 column (tid&31)
    // Named registers (not arrays): hipcc keeps small float4 arrays that are
    // written under a loop-carried condition in scratch memory.
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    const int a_r = tid >> 3, a_c = (tid & 7) * 4;
    const int b_r = tid >> 5, b_c = (tid & 31) * 4;
    int64_t a_off[4], b_off[4];   // FAST path: per-thread element offsets of the 8 staged float4s
    if constexpr (FAST) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = m0 + a_r + 32 * i;
            row = row < g.M ? row : g.M - 1;           // clamp: loads stay in bounds, stores are guarded
            a_off[i] = (int64_t)row * g.lda + a_c;
            int col = n0 + b_c;
            col = col < g.N ? col : g.N - 4;           // clamped columns feed accumulators that are never stored
            b_off[i] = (int64_t)(b_r + 8 * i) * g.ldb + col;
        }
    }

#define W2V2_LD_A(i_, k0_)                                                                         \
    (FAST ? *reinterpret_cast<const float4*>(A + a_off[i_] + (k0_))                                \
          : make_float4(ld_a(A, g, m0 + a_r + 32 * (i_), (k0_) + a_c),                             \
                        ld_a(A, g, m0 + a_r + 32 * (i_), (k0_) + a_c + 1),                         \
                        ld_a(A, g, m0 + a_r + 32 * (i_), (k0_) + a_c + 2),                         \
                        ld_a(A, g, m0 + a_r + 32 * (i_), (k0_) + a_c + 3)))
#define W2V2_LD_B(i_, k0_)                                                                         \
    (FAST ? *reinterpret_cast<const float4*>(Bm + b_off[i_] + (int64_t)(k0_) * g.ldb)              \
          : make_float4(ld_b(Bm, g, (k0_) + b_r + 8 * (i_), n0 + b_c),                             \
                        ld_b(Bm, g, (k0_) + b_r + 8 * (i_), n0 + b_c + 1),                         \
                        ld_b(Bm, g, (k0_) + b_r + 8 * (i_), n0 + b_c + 2),                         \
                        ld_b(Bm, g, (k0_) + b_r + 8 * (i_), n0 + b_c + 3)))
#define W2V2_LOAD_TILE(kt_)                                                                        \
    do {                                                                                           \
        const int k0__ = (kt_) * BK;                                                               \
        ra0 = W2V2_LD_A(0, k0__); ra1 = W2V2_LD_A(1, k0__);                                        \
        ra2 = W2V2_LD_A(2, k0__); ra3 = W2V2_LD_A(3, k0__);                                        \
        rb0 = W2V2_LD_B(0, k0__); rb1 = W2V2_LD_B(1, k0__);                                        \
        rb2 = W2V2_LD_B(2, k0__); rb3 = W2V2_LD_B(3, k0__);                                        \
    } while (0)
#define W2V2_STORE_TILE(buf_)                                                                      \
    do {                                                                                           \
        float* As_ = smem + (buf_) * STAGE_FLOATS + a_r * LDA_S + a_c;                             \
        float* Bs_ = smem + (buf_) * STAGE_FLOATS + BM * LDA_S + b_r * LDB_S + b_c;                \
        *reinterpret_cast<float4*>(As_) = ra0;                                                     \
        *reinterpret_cast<float4*>(As_ + 32 * LDA_S) = ra1;                                        \
        *reinterpret_cast<float4*>(As_ + 64 * LDA_S) = ra2;                                        \
        *reinterpret_cast<float4*>(As_ + 96 * LDA_S) = ra3;                                        \
        *reinterpret_cast<float4*>(Bs_) = rb0;                                                     \
        *reinterpret_cast<float4*>(Bs_ + 8 * LDB_S) = rb1;                                         \
        *reinterpret_cast<float4*>(Bs_ + 16 * LDB_S) = rb2;                                        \
        *reinterpret_cast<float4*>(Bs_ + 24 * LDB_S) = rb3;                                        \
    } while (0)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = (g.K + BK - 1) / BK;
    W2V2_LOAD_TILE(0);
    W2V2_STORE_TILE(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) W2V2_LOAD_TILE(kt + 1);     // in flight under the MFMAs below

        const float* As = smem + cur * STAGE_FLOATS + (wm * 64 + li) * LDA_S + 4 * lh;
        const float* Bs = smem + cur * STAGE_FLOATS + BM * LDA_S + (4 * lh) * LDB_S + wn * 64 + li;
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            float4 a[2];
            float b[2][4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                a[mt] = *reinterpret_cast<const float4*>(As + mt * 32 * LDA_S + kb * 8);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) b[nt][e] = Bs[(kb * 8 + e) * LDB_S + nt * 32];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const float av = e == 0 ? a[mt].x : e == 1 ? a[mt].y : e == 2 ? a[mt].z : a[mt].w;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[nt][e], acc[mt][nt], 0, 0, 0);
                }
            }
        }
        if (kt + 1 < nk) W2V2_STORE_TILE(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias -> activation -> + residual -> store -------------------
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float* __restrict__ C = g.C + (int64_t)z * g.strideC;
    const float* __restrict__ R = g.residual ? g.residual + (int64_t)z * g.strideC : nullptr;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = n0 + wn * 64 + nt * 32 + li;
        if (col >= g.N) continue;
        const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < g.M) {
                    float v = apply_act(acc[mt][nt][r] + bv, g.act);
                    if (R) v += R[(int64_t)row * g.ldc + col];
                    C[(int64_t)row * g.ldc + col] = v;
                }
            }
        }
    }
}

}  // namespace

int launch_gemm(Profiler* prof, const float* A, int64_t lda, int64_t strideA, const float* B,
                int64_t ldb, float* C, int64_t ldc, int64_t strideC, const float* bias,
                const float* residual, int M, int N, int K, int nbatch, int act, hipStream_t s) {
    W2V2_REQUIRE(A && B && C, "gemm: null operand");
    W2V2_REQUIRE(M > 0 && N > 0 && K > 0 && nbatch > 0, "gemm: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, nbatch);
    W2V2_REQUIRE(lda >= 1 && ldb >= N && ldc >= N, "gemm: bad leading dimensions");
    W2V2_REQUIRE(act >= 0 && act <= 2, "gemm: bad activation %d", act);
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.residual = residual;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.strideA = strideA; g.strideC = strideC;
    g.M = M; g.N = N; g.K = K; g.act = act;
    g.tiles_m = (M + BM - 1) / BM;
    g.tiles_n = (N + BN - 1) / BN;
    const bool fast = (K % BK == 0) && (N % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                      (strideA % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    dim3 grid(g.tiles_m * g.tiles_n, 1, nbatch), block(256);
    const size_t lds = 2 * STAGE_FLOATS * sizeof(float);   // 68 KiB: above the 64 KiB default cap
    static bool attr_set = false;
    if (!attr_set) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    ProfScope ps(prof, FAM_GEMM, 2.0 * M * (double)N * K * nbatch,
                 4.0 * nbatch * ((double)M * K + (double)M * N) + 4.0 * (double)K * N, s);
    if (fast)
        hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, block, lds, s, g);
    else
        hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, block, lds, s, g);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace w2v2
