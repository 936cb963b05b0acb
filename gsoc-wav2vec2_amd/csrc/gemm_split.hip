// fp32 GEMM on the bf16 matrix cores ("bf16x3" split, precision mode 2).
//
// gfx950 multiplies bf16 16x faster than fp32 (2.5 PFLOP/s against 157 TFLOP/s dense), so an fp32 product is cheaper
// as several bf16 products than as one fp32 MFMA.  Every fp32 operand is written EXACTLY as the sum of three bf16
// numbers,  x = x0 + x1 + x2  with  x0 = bf16(x), x1 = bf16(x - x0), x2 = x - x0 - x1  (24 significand bits = 3 x 8;
// the two subtractions are exact in fp32, the last remainder fits 8 bits), and the contraction keeps the six
// products of order <= 2:
//     a b  ~=  a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0),       dropped: a1 b2 + a2 b1 + a2 b2 <= 2^-23 |a b|
// Each bf16 x bf16 product is exact in fp32 and the MFMA accumulates in fp32, smallest terms first.  The dropped part
// is below one ulp of the fp32 product, i.e. below the rounding an fp32 running sum commits on every step anyway:
// tests/test_ops_gpu.py measures the error against fp64 next to the native fp32 MFMA kernel's (same order of magnitude;
// the model-level logits keep the fp32 path's distance to the fp64 reference).
//
// Dataflow per 128 x 256 x 16 tile step (8 waves, each a 64 x 64 block of 32x32 accumulators; BK = 16 is the default,
// a BK = 32 instantiation is kept behind W2V2_SPLIT_BK for comparison):
//   * B (a weight, constant across calls) is pre-split once into three bf16 planes stored as this kernel's LDS images,
//     [K / BK][plane][N][BK] with the slot XOR already applied (launch_split_weight); each step's 3 x 256 x 16 slab is
//     24 KiB of consecutive memory that global_load_lds_dwordx4 copies HBM/L2 -> LDS with linear addresses (no registers,
//     no VALU, every cache line used once),
//   * A (an activation, fp32 in HBM) is loaded as float4, split in registers (11 VALU ops per 4 elements) and written
//     to three LDS planes,
//   * 12 ds_read_b128 feed 24 MFMAs per wave and k16 step.
// LDS rows are 2 BK bytes; the 16-byte slot index is XOR-ed with row bits ((row >> 3) & 1 at BK = 16, (row >> 2) & 3 at
// BK = 32), which makes the fragment reads and the A stores bank-conflict free (SQ_LDS_BANK_CONFLICT = 0).
// LDS: 2 stages x 3 x (128 + 256) x 32 B = 72 KiB, two 512-thread blocks per CU (4 waves per SIMD, <= 128 VGPRs).
// The staging of the next step is threaded between the MFMA groups (see `step` below); results do not depend on the
// tiling: every output element sums its K products in the same order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"

namespace w2v2 {

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

constexpr int BM = 128, BN = 256, WM = 2, WN = 4, NT = WM * WN * 64;
constexpr int MT = 2, NTL = 2;                             // 64 x 64 wave tile

struct SplitArgs {
    const float* A;
    const uint16_t* Bp;       // the weight as LDS images: [K / BK][plane][N][BK] bf16 (launch_split_weight)
    float* C;
    const float* bias;
    const float* residual;
    int64_t lda, ldc, strideA, strideC;
    int M, N, K, act;
    int tiles_m, tiles_n, order;
};

// x (4 floats) -> three dword pairs: planes 0..2, each 4 bf16
__device__ __forceinline__ void split4(const f32x4& x, u32x2& p0, u32x2& p1, u32x2& p2) {
    p0[0] = pack_bf16_rne(x[0], x[1]);
    p0[1] = pack_bf16_rne(x[2], x[3]);
    f32x4 r;
    r[0] = x[0] - __uint_as_float(p0[0] << 16);
    r[1] = x[1] - __uint_as_float(p0[0] & 0xffff0000u);
    r[2] = x[2] - __uint_as_float(p0[1] << 16);
    r[3] = x[3] - __uint_as_float(p0[1] & 0xffff0000u);
    p1[0] = pack_bf16_rne(r[0], r[1]);
    p1[1] = pack_bf16_rne(r[2], r[3]);
    r[0] -= __uint_as_float(p1[0] << 16);
    r[1] -= __uint_as_float(p1[0] & 0xffff0000u);
    r[2] -= __uint_as_float(p1[1] << 16);
    r[3] -= __uint_as_float(p1[1] & 0xffff0000u);
    p2[0] = pack_bf16_rne(r[0], r[1]);
    p2[1] = pack_bf16_rne(r[2], r[3]);
}

// BK = k extent of one LDS stage: 32 (64-byte rows, 144 KiB, one block per CU) or 16 (32-byte rows, 72 KiB, two blocks per CU)
template <int BK>
struct SplitCfg {
    static constexpr int ROWB = BK * 2;                              // bytes per LDS row
    static constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
    static constexpr int STAGE = 3 * A_PLANE + 3 * B_PLANE;
    static constexpr int NA = BM * BK / 4 / NT;                      // float4 of the A slab per thread
    static constexpr int LPR = BK / 4;                               // lanes (float4) per A row
    static constexpr int PROWS = 1024 / ROWB;                        // rows per 1-KiB DMA piece
    static constexpr int NPB = 3 * BN / PROWS / (NT / 64);           // DMA pieces per wave
    static constexpr int SUB = BK / 16;                              // k16 sub-steps per stage
    // 16-byte slot XOR: consecutive rows walk the 256-byte bank span once, then the slot index changes
    static constexpr __device__ __host__ __forceinline__ int sw(int row) { return BK == 32 ? (row >> 2) & 3 : (row >> 3) & 1; }
};

template <int BK, int MINW>   // MINW: waves per SIMD the register budget must allow (HIP's second launch-bound)
__global__ __launch_bounds__(NT, MINW) void gemm_split_kernel(SplitArgs g) {
    using Cf = SplitCfg<BK>;
    constexpr int ROWB = Cf::ROWB, A_PLANE = Cf::A_PLANE, B_PLANE = Cf::B_PLANE, STAGE = Cf::STAGE, NA = Cf::NA, LPR = Cf::LPR,
                  PROWS = Cf::PROWS, NPB = Cf::NPB, SUB = Cf::SUB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_split[];
    const int tid = threadIdx.x, lane = tid & 63;
    // (readfirstlane: tells the compiler the wave index is uniform, so the B pieces' sources become scalar bases + one lane offset
    //  instead of three 64-bit pointers per lane -- this kernel sits on its 128-VGPR limit: forward 42.3 -> 39.4 ms with this and the
    //  single fragment offset below)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;

    // XCD-aware tile order (gemm_f32.hip): each XCD walks a contiguous run of tiles, N fastest
    const int nwg = g.tiles_m * g.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = g.order ? bid % g.tiles_m : bid / g.tiles_n, tn = g.order ? bid / g.tiles_m : bid % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.z;
    const float* __restrict__ A = g.A + (int64_t)z * g.strideA;
    const int nk = g.K / BK;

    // ---- A: BM rows x BK k fp32, NA float4 per thread (LPR lanes cover one row segment) ----
    f32x4 ra[NA];
    const float* a_src[NA];
    int a_lds[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = tid / LPR + (NT / LPR) * i, q = tid % LPR;
        int row = m0 + r;
        row = row < g.M ? row : g.M - 1;              // clamped rows feed accumulators that are never stored
        a_src[i] = A + (int64_t)row * g.lda + q * 4;
        a_lds[i] = r * ROWB + (((q >> 1) ^ Cf::sw(r)) << 4) + (q & 1) * 8;
    }
    // ---- B: 3 planes x BN rows x ROWB bytes in 1-KiB pieces (PROWS rows), NPB per wave ----
    const uint16_t* b_src[NPB];
    int b_lds[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        constexpr int PPP = BN / PROWS;               // pieces per plane
        const int piece = wave * NPB + i, plane = piece / PPP, rb = (piece % PPP) * PROWS;
        // the planes are stored as LDS images (launch_split_weight): [k tile][plane][n][BK] with the slot XOR applied,
        // so a piece is 1 KiB of consecutive memory and the per-step advance is 3 N BK elements
        b_src[i] = g.Bp + ((int64_t)plane * g.N + n0 + rb) * BK;      // wave-uniform; the lane's 16 bytes are added at the issue
        b_lds[i] = 3 * A_PLANE + plane * B_PLANE + rb * ROWB;
    }
    const int64_t b_step = (int64_t)3 * g.N * BK;
    auto load_a = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(a_src[i] + kt * BK);
    };
    auto issue_piece = [&](int i, int kt, int buf) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + (int64_t)kt * b_step + lane * 8),
                                         (__attribute__((address_space(3))) void*)(smem_split + buf * STAGE + b_lds[i]), 16, 0, 0);
    };
    auto store_a = [&](int buf) {
        unsigned char* S = smem_split + buf * STAGE;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            u32x2 p0, p1, p2;
            split4(ra[i], p0, p1, p2);
            *reinterpret_cast<u32x2*>(S + a_lds[i]) = p0;
            *reinterpret_cast<u32x2*>(S + A_PLANE + a_lds[i]) = p1;
            *reinterpret_cast<u32x2*>(S + 2 * A_PLANE + a_lds[i]) = p2;
        }
    };

    f32x16_t acc[MT][NTL];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    // fragment byte offsets inside a plane for k16 sub-step 0 (sub-step 1: slot ^ 2).  The swizzle has period 16 in the row, so
    // the second 32-row fragment of a wave sits exactly 32 rows further: ONE per-lane offset per operand, the rest are immediates
    // (two registers fewer in a kernel that sits on its 128-VGPR limit)
    static_assert(Cf::sw(0) == Cf::sw(32) && Cf::sw(5) == Cf::sw(37) && Cf::sw(12) == Cf::sw(44), "swizzle period must divide 32");
    const int fa0 = (wm * 64 + li) * ROWB + ((lh ^ Cf::sw(wm * 64 + li)) << 4);
    const int fb0 = 3 * A_PLANE + (wn * 64 + li) * ROWB + ((lh ^ Cf::sw(wn * 64 + li)) << 4);
    int fa[MT], fb[NTL];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) fa[mt] = fa0 + mt * 32 * ROWB;
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) fb[nt] = fb0 + nt * 32 * ROWB;
    // One tile step = 6 SUB groups of four MFMAs on stage `buf`.  The staging of the next step is threaded BETWEEN the
    // groups instead of bunched at the stage boundary (where all eight waves would stall on the same unit together and
    // leave the matrix pipe idle): group 0 is followed by the split + LDS store of the A slab already in registers,
    // groups 1..NPB by one LDS-DMA piece of B each (an LDS-DMA issue holds its wave for 60-180 cycles), the next group
    // by the global load of the A slab after that.
    auto step = [&](int buf, int kt, auto prefetch) {
        constexpr bool PF = decltype(prefetch)::value;
        const unsigned char* S = smem_split + buf * STAGE;
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
            bf16x8 a[MT][3], b[NTL][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt][p] = *reinterpret_cast<const bf16x8*>(S + p * A_PLANE + (fa[mt] ^ (s << 5)));
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) b[nt][p] = *reinterpret_cast<const bf16x8*>(S + p * B_PLANE + (fb[nt] ^ (s << 5)));
            }
            // smallest terms first: (a2 b0, a0 b2, a1 b1), (a1 b0, a0 b1), a0 b0; four independent accumulators per term
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTL; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][PA[t]], b[nt][PB[t]], acc[mt][nt], 0, 0, 0);
                if constexpr (PF) {
                    const int grp = s * 6 + t;
                    __builtin_amdgcn_sched_barrier(0);
                    if (grp == 0) store_a(buf ^ 1);
                    else if (grp <= NPB) issue_piece(grp - 1, kt + 1, buf ^ 1);
                    else if (grp == NPB + 1) load_a(kt + 2 < nk ? kt + 2 : nk - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    static_assert(NPB + 1 < 6 * SUB, "not enough MFMA groups to carry the staging");
    auto stage_fence = [&]() {       // this wave's DMA pieces have landed (the NA younger A loads may still fly), LDS stores done
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    load_a(0);
#pragma unroll
    for (int i = 0; i < NPB; ++i) issue_piece(i, 0, 0);
    store_a(0);
    __builtin_amdgcn_sched_barrier(0);
    load_a(nk > 1 ? 1 : 0);
    stage_fence();
    for (int kt = 0; kt + 1 < nk; ++kt) {
        step(kt & 1, kt, std::true_type{});
        stage_fence();
    }
    step((nk - 1) & 1, nk - 1, std::false_type{});

    const int64_t tile_off = (int64_t)z * g.strideC + (int64_t)(m0 + wm * 64) * g.ldc + (n0 + wn * 64);
    gemm_epilogue<MT, NTL, false>(acc, g.C + tile_off, nullptr, g.residual ? g.residual + tile_off : nullptr,
                                 g.bias ? g.bias + (n0 + wn * 64) : nullptr, (int)g.ldc, g.M - (m0 + wm * 64),
                                 g.N - (n0 + wn * 64), g.act, li, lh);
}

// w (K, N) row-major fp32  ->  the kernel's LDS images: [K / BK][plane 0..2][N][BK] bf16, 16-byte slots XOR-ed as in LDS
template <int BK>
__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ planes, int K, int N) {
    __shared__ float tile[64][65];
    const int k0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int k = k0 + r, n = n0 + tx;
        tile[r][tx] = (k < K && n < N) ? w[(int64_t)k * N + n] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int n = n0 + r, k = k0 + tx;
        if (n < N && k < K) {
            const float x = tile[tx][r];
            const unsigned h0 = pack_bf16_rne(x, 0.f) & 0xffffu;
            const float r1 = x - __uint_as_float(h0 << 16);
            const unsigned h1 = pack_bf16_rne(r1, 0.f) & 0xffffu;
            const float r2 = r1 - __uint_as_float(h1 << 16);
            const unsigned h2 = pack_bf16_rne(r2, 0.f) & 0xffffu;
            const int kt = k / BK, kk = k % BK;
            const int64_t o = ((int64_t)kt * 3 * N + n) * BK + (((kk >> 3) ^ SplitCfg<BK>::sw(n & (BN - 1))) << 3) + (kk & 7);
            const int64_t plane = (int64_t)N * BK;
            planes[o] = (uint16_t)h0;
            planes[o + plane] = (uint16_t)h1;
            planes[o + 2 * plane] = (uint16_t)h2;
        }
    }
}

// k extent of one LDS stage; fixed for the process because the weight planes are stored in that tiling
int split_bk() {
    static int bk = -1;
    if (bk < 0) bk = tune_int("W2V2_SPLIT_BK", 16) == 32 ? 32 : 16;
    return bk;
}

}  // namespace

bool gemm_split_supported(const float* A, int64_t lda, int64_t strideA, int M, int N, int K) {
    return M > 0 && N % BN == 0 && K % 32 == 0 && lda % 4 == 0 && strideA % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(A) & 15) == 0;
}

int launch_split_weight(const float* w, uint16_t* planes, int K, int N, hipStream_t s) {
    W2V2_REQUIRE(w && planes && K > 0 && N > 0, "split_weight: bad argument");
    W2V2_REQUIRE(K % 32 == 0 && N % BN == 0, "split_weight: needs K %% 32 == 0 and N %% 256 == 0");
    if (split_bk() == 32)
        W2V2_LAUNCH(split_weight_kernel<32>, dim3((N + 63) / 64, (K + 63) / 64), dim3(256), 0, s, w, planes, K, N);
    else
        W2V2_LAUNCH(split_weight_kernel<16>, dim3((N + 63) / 64, (K + 63) / 64), dim3(256), 0, s, w, planes, K, N);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_gemm_split(Profiler* prof, const float* A, int64_t lda, int64_t strideA, const uint16_t* planes, float* C, int64_t ldc,
                      int64_t strideC, const float* bias, const float* residual, int M, int N, int K, int nbatch, int act,
                      hipStream_t s) {
    W2V2_REQUIRE(A && planes && C && nbatch > 0, "gemm_split: null operand");
    W2V2_REQUIRE(gemm_split_supported(A, lda, strideA, M, N, K), "gemm_split: needs N %% 256 == 0, K %% 32 == 0, 16-byte aligned A rows");
    W2V2_REQUIRE(ldc >= N && ldc < (1 << 23) && act >= 0 && act <= 2, "gemm_split: bad leading dimension / activation");
    W2V2_REQUIRE((reinterpret_cast<uintptr_t>(planes) & 15) == 0, "gemm_split: unaligned weight planes");
    SplitArgs g;
    g.A = A; g.Bp = planes; g.C = C; g.bias = bias; g.residual = residual;
    g.lda = lda; g.ldc = ldc; g.strideA = strideA; g.strideC = strideC;
    g.M = M; g.N = N; g.K = K; g.act = act;
    g.tiles_m = (M + BM - 1) / BM;
    g.tiles_n = N / BN;
    ProfScope ps(prof, FAM_GEMM_SPLIT, 2.0 * M * (double)N * K * nbatch,
                 nbatch * 4.0 * ((double)M * K + (double)M * N) + 6.0 * (double)K * N, s);
    static int order = -1;
    if (order < 0) order = tune_int("W2V2_SPLIT_ORDER", 1);
    g.order = order;
    const int bk = split_bk();
    const dim3 grid(g.tiles_m * g.tiles_n, 1, nbatch);
    if (bk == 32) {
        constexpr size_t LDS = 2 * SplitCfg<32>::STAGE;
        static std::atomic<bool> attr_set{false};   // (idempotent call; atomic so concurrent host threads agree on the flag)
        if (!attr_set) {
            W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<32, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
            attr_set = true;
        }
        W2V2_LAUNCH((gemm_split_kernel<32, 2>), grid, dim3(NT), LDS, s, g);
    } else {
        constexpr size_t LDS = 2 * SplitCfg<16>::STAGE;
        static std::atomic<bool> attr_set{false};   // (idempotent call; atomic so concurrent host threads agree on the flag)
        if (!attr_set) {
            W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
            attr_set = true;
        }
        W2V2_LAUNCH((gemm_split_kernel<16, 4>), grid, dim3(NT), LDS, s, g);
    }
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace w2v2
