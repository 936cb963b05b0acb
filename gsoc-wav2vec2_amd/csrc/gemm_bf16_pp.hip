// bf16 GEMM, 256 x 256 x 64 tiles, two wave groups in ping-pong, operand half-tiles through an eight-slot LDS ring.
//
// Why a second kernel (profiles/r03_gemm_bf16_study.md): per-block phase traces of the 128 x 128 kernel (gemm_bf16.hip) show its
// K loop bound by the operand stream, not by the matrix pipe -- a 128 x 128 x 64 step moves 32 KiB from L2 into LDS for 512
// matrix-pipe cycles, which is the CU's whole L2 -> LDS rate (~56-64 B/clk), and with one tile of prefetch every step also waits
// out one L2 round trip (1.3-2.6 k cycles per step against 1 k of MFMA work for the two resident blocks).  A 256 x 256 tile halves
// the bytes per flop, and a ring of half-tiles keeps up to six 16-KiB pieces of the stream in flight per block.
//
// Structure (one block per CU, 8 waves = 2 (M) x 4 (N), a wave owns 128 x 64 outputs = 4 x 2 accumulators of 32 x 32):
//   * A K tile (64 k) of the block's operands is four HALF-TILES of 16 KiB: A_0 / A_1 = the first / second 64 rows of each wave
//     row group's 128 rows, B_0 / B_1 = the first / second 32 columns of each wave column group's 64.  LDS holds two K tiles =
//     eight slots.  The stream order is A_0(0) | B_0 B_1 A_1 A_0(next) | ... one half-tile per PHASE.
//   * A phase = [load segment: this wave's ds_reads of ONE half-tile into registers (4 or 8 ds_read_b128) + its two 1-KiB
//     LDS-DMA pieces of the half-tile D phases ahead] s_waitcnt vmcnt(2 (D - 1)) / s_barrier / lgkmcnt(0) / 8 MFMAs (one
//     64 x 32 quadrant of the wave's tile over the whole K tile) / s_barrier.  Quadrant order (0,0) (0,1) (1,1) (1,0): each phase
//     needs exactly one new half-tile, every half-tile is read from LDS once per wave, and its slot is free D <= 6 phases before
//     the stream comes round to it again.
//   * The two wave groups (M halves; the two waves of every SIMD are one of each) run one barrier apart, so while one wave of a
//     SIMD issues its 8 MFMAs the other is in its load segment: the matrix pipe always has a wave to serve.
//   * Waits are counted, never zero in steady state; the reads of a half-tile come one barrier after every wave has waited for its
//     own pieces of it (LDS-DMA is ordered for a ds_read only by the issuing wave's vmcnt plus a barrier the reader has passed).
//
// Same contract and the same arithmetic as gemm_bf16_kernel's shadow-fed form: every output sums its k products in ascending
// 16-deep MFMA steps with the same lane -> k assignment, so the two kernels give identical bits (tests/test_ops_gpu.py).
#include "common.h"
#include "gemm_epilogue.h"

namespace w2v2 {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int PP_BM = 256, PP_BN = 256, PP_BK = 64;
constexpr int PP_SLOT = 16384;                    // one half-tile: 128 image rows x 128 B
constexpr int PP_LDS = 8 * PP_SLOT;               // two K tiles

struct GemmPPArgs {
    const uint16_t* A16;
    const uint16_t* B16;       // (N, K) rows ldb16 apart
    float* C;
    uint16_t* C16;
    const float* bias;
    const float* residual;
    int64_t lda, ldb16, ldc, strideA, strideC;
    int M, N, K, act;
    int tiles_m, tiles_n;
#ifdef W2V2_TUNING
    unsigned long long* trace;
#endif
};

__device__ __forceinline__ int pp_swz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 1); }      // = gemm_bf16.hip's swz

template <int OFF>
__device__ __forceinline__ bf16x8 pp_read(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

template <int N>
__device__ __forceinline__ void pp_wait_vm() {
    if constexpr (N >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int V>
using IC = std::integral_constant<int, V>;

// half-tile type of stream item s: 0 = A_0, 1 = B_0, 2 = B_1, 3 = A_1;  LDS slot of that type inside a K-tile buffer
__host__ __device__ constexpr int pp_slot_of_type(int type) { return type == 0 ? 0 : type == 1 ? 2 : type == 2 ? 3 : 1; }

template <int D, bool PRIO, bool TRACE = false>
__global__ __launch_bounds__(512, 1) void gemm_bf16_pp_kernel(GemmPPArgs g) {
    static_assert(D >= 2 && D <= 6, "prefetch distance: a slot is rewritten 8 phases after it was read, minus the group stagger");
    extern __shared__ __attribute__((aligned(16))) unsigned char pp_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3, li = lane & 31, lh = lane >> 5;

    // XCD-aware tile order (gemm_f32.hip): each XCD walks a contiguous run of tiles, N fastest
    const int nwg = g.tiles_m * g.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / g.tiles_n, tn = bid % g.tiles_n;
    const int m0 = tm * PP_BM, n0 = tn * PP_BN;
    const int z = blockIdx.z;
    const int nk = g.K / PP_BK;

#ifdef W2V2_TUNING
    unsigned long long* const trc = (TRACE && g.trace) ? g.trace + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 32 : nullptr;
    int trc_n = 2;
    if (TRACE && trc && tid == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trc[0] = ((unsigned long long)xcc << 32) | hwid;
        trc[1] = wall_clock64();
        trc[trc_n++] = clock64();
    }
#define PP_TRC() do { if (TRACE && trc && tid == 0 && trc_n < 31) trc[trc_n++] = clock64(); } while (0)
#else
#define PP_TRC() do { } while (0)
#endif

    // ---- LDS-DMA sources: per half-tile type two 1-KiB pieces per wave; piece pc = rows 8 pc .. 8 pc + 7 of the 128-row image,
    // the lane at physical 16-byte slot (lane & 7) of image row r fetches logical slot (lane & 7) ^ swz(r).  Offsets are bytes
    // from the tile's A / B base (rows clamped to the matrix; < 256 ld x 2 < 2^31), the K-tile advance is scalar.
    uint32_t doff[4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + (lane >> 3);                    // image row 0 .. 127
        const uint32_t sl = (uint32_t)(((lane & 7) ^ pp_swz(r)) << 3);     // logical slot, in elements
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int ar = (r >> 6) * 128 + h * 64 + (r & 63);                   // tile row of A_h's image row r
            ar = m0 + ar < g.M ? ar : g.M - 1 - m0;
            int bc = (r >> 5) * 64 + h * 32 + (r & 31);                    // tile column of B_h's image row r
            bc = n0 + bc < g.N ? bc : g.N - 1 - n0;
            doff[h == 0 ? 0 : 3][i] = 2u * ((uint32_t)((int64_t)ar * g.lda) + sl);
            doff[h == 0 ? 1 : 2][i] = 2u * ((uint32_t)((int64_t)bc * g.ldb16) + sl);
        }
    }
    const uint16_t* const baseA = g.A16 + (int64_t)z * g.strideA + (int64_t)m0 * g.lda;
    const uint16_t* const baseB = g.B16 + (int64_t)n0 * g.ldb16;
    auto uniform_ptr = [](const uint16_t* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const unsigned char*>(((uint64_t)hi << 32) | lo);
    };
    // stream item of type TYPE belonging to K tile `ktile` (parity PAR = ktile & 1 known at compile time)
    auto issue = [&](auto TYPEc, auto PARc, int ktile) {
        constexpr int TYPE = decltype(TYPEc)::value, PAR = decltype(PARc)::value;
        constexpr bool ISA = TYPE == 0 || TYPE == 3;
        unsigned char* S = pp_smem + (PAR * 4 + pp_slot_of_type(TYPE)) * PP_SLOT;
        const unsigned char* const u = uniform_ptr((ISA ? baseA : baseB) + ktile * PP_BK);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(u + doff[TYPE][i]),
                                             (__attribute__((address_space(3))) void*)(S + (wave * 2 + i) * 1024), 16, 0, 0);
    };

    // ---- fragment reads: image row rho (A: wr 64 + rb 32 + li; B: wc 32 + li), logical slot 2 ks + lh -> physical ^ swz(rho).
    // (2 ks + lh) ^ sw = ((2 ks) ^ (sw & 6)) + (lh ^ (sw & 1)): a per-lane constant plus an XOR of the k step with a per-lane mask.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)pp_smem;
    const int rhoA = wr * 64 + li, rhoB = wc * 32 + li;
    const unsigned xA = lds0 + (unsigned)rhoA * 128u + 16u * (unsigned)(lh ^ (pp_swz(rhoA) & 1)), yA = (unsigned)(pp_swz(rhoA) & 6) * 16u;
    const unsigned xB = lds0 + (unsigned)rhoB * 128u + 16u * (unsigned)(lh ^ (pp_swz(rhoB) & 1)), yB = (unsigned)(pp_swz(rhoB) & 6) * 16u;

    bf16x8 fa[2][2][4];      // [i: 64-row half][rb: 32-row block][ks]
    bf16x8 fb[2][4];         // [j: 32-column half][ks]
    auto read_a = [&](auto Ic, auto PARc) {
        constexpr int I = decltype(Ic)::value, PAR = decltype(PARc)::value;
        const unsigned base = xA + (unsigned)(PAR * 4 * PP_SLOT);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned a = base + ((32u * ks) ^ yA);
            fa[I][0][ks] = pp_read<I * PP_SLOT>(a);
            fa[I][1][ks] = pp_read<I * PP_SLOT + 4096>(a);
        }
    };
    auto read_b = [&](auto Jc, auto PARc) {
        constexpr int J = decltype(Jc)::value, PAR = decltype(PARc)::value;
        const unsigned base = xB + (unsigned)(PAR * 4 * PP_SLOT);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb[J][ks] = pp_read<(2 + J) * PP_SLOT>(base + ((32u * ks) ^ yB));
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // One phase.  Q = position in the K tile; PAR = parity of the K tile `kt`; ISSUE: this phase still has a half-tile to request;
    // VM = vmcnt to wait for before the barrier (-1: nothing to wait for); READ: the half-tile of this phase's load segment exists.
    auto phase = [&](auto Qc, auto PARc, auto ISSUEc, auto VMc, auto READc, int kt) {
        constexpr int Q = decltype(Qc)::value, PAR = decltype(PARc)::value, VM = decltype(VMc)::value;
        constexpr bool ISSUE = decltype(ISSUEc)::value != 0, READ = decltype(READc)::value != 0;
        // ---- load segment: stream item 4 kt + Q + 1
        if constexpr (READ) {
            if constexpr (Q == 0) read_b(IC<0>{}, IC<PAR>{});
            if constexpr (Q == 1) read_b(IC<1>{}, IC<PAR>{});
            if constexpr (Q == 2) read_a(IC<1>{}, IC<PAR>{});
            if constexpr (Q == 3) read_a(IC<0>{}, IC<PAR ^ 1>{});        // A_0 of the NEXT K tile
        }
        if constexpr (ISSUE) {
            constexpr int S = Q + 1 + D;                                  // item 4 kt + S: type S & 3 of K tile kt + (S >> 2)
            issue(IC<(S & 3)>{}, IC<(PAR ^ ((S >> 2) & 1))>{}, kt + (S >> 2));
        }
        __builtin_amdgcn_sched_barrier(0);
        pp_wait_vm<VM>();
        __builtin_amdgcn_s_barrier();
        // ---- compute segment: quadrant (i, j) of the wave's tile over the whole K tile
        constexpr int QI = (Q == 0 || Q == 1) ? 0 : 1, QJ = (Q == 0 || Q == 3) ? 0 : 1;
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fa[QI][0][0]), "+v"(fa[QI][0][1]), "+v"(fa[QI][0][2]), "+v"(fa[QI][0][3]), "+v"(fa[QI][1][0]), "+v"(fa[QI][1][1]),
                       "+v"(fa[QI][1][2]), "+v"(fa[QI][1][3]), "+v"(fb[QJ][0]), "+v"(fb[QJ][1]), "+v"(fb[QJ][2]), "+v"(fb[QJ][3]));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                acc[QI * 2 + rb][QJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[QI][rb][ks], fb[QJ][ks], acc[QI * 2 + rb][QJ], 0, 0, 0);
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    // ---- prologue: stream items 0 .. D in flight, items 0 (A_0 of K tile 0) and 1 landed, A_0 in registers
    {
        auto pro = [&](auto Sc) {
            constexpr int S = decltype(Sc)::value;
            if constexpr (S <= D) issue(IC<(S & 3)>{}, IC<((S >> 2) & 1)>{}, S >> 2);
        };
        pro(IC<0>{}); pro(IC<1>{}); pro(IC<2>{}); pro(IC<3>{}); pro(IC<4>{}); pro(IC<5>{}); pro(IC<6>{});
        pp_wait_vm<2 * (D - 1)>();
        __builtin_amdgcn_s_barrier();
        PP_TRC();
        read_a(IC<0>{}, IC<0>{});
        if (wr == 1) __builtin_amdgcn_s_barrier();      // the second group runs one barrier behind from here on
    }

    constexpr int VMS = 2 * (D - 1);                     // steady state: D - 1 younger half-tiles may stay in flight
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
        phase(IC<0>{}, IC<0>{}, IC<1>{}, IC<VMS>{}, IC<1>{}, kt);
        phase(IC<1>{}, IC<0>{}, IC<1>{}, IC<VMS>{}, IC<1>{}, kt);
        phase(IC<2>{}, IC<0>{}, IC<1>{}, IC<VMS>{}, IC<1>{}, kt);
        phase(IC<3>{}, IC<0>{}, IC<1>{}, IC<VMS>{}, IC<1>{}, kt);
        phase(IC<0>{}, IC<1>{}, IC<1>{}, IC<VMS>{}, IC<1>{}, kt + 1);
        phase(IC<1>{}, IC<1>{}, IC<1>{}, IC<VMS>{}, IC<1>{}, kt + 1);
        phase(IC<2>{}, IC<1>{}, IC<1>{}, IC<VMS>{}, IC<1>{}, kt + 1);
        phase(IC<3>{}, IC<1>{}, IC<1>{}, IC<VMS>{}, IC<1>{}, kt + 1);
    }
    PP_TRC();
    // ---- the last two K tiles: phases t = 7 .. 0 from the end.  A phase requests a half-tile while t >= D + 1; before its barrier
    // the half-tile of the NEXT phase's load segment must have landed: min(D - 1, t - 2) younger ones may stay in flight (t >= 2).
    {
#define PP_TAIL(T, Q, PAR, KT)                                                                                                        \
    phase(IC<Q>{}, IC<PAR>{}, IC<((T) >= D + 1)>{}, IC<((T) >= 2 ? 2 * ((D - 1) < (T)-2 ? (D - 1) : (T)-2) : -1)>{}, IC<((T) >= 1)>{}, KT)
        PP_TAIL(7, 0, 0, kt);
        PP_TAIL(6, 1, 0, kt);
        PP_TAIL(5, 2, 0, kt);
        PP_TAIL(4, 3, 0, kt);
        PP_TAIL(3, 0, 1, kt + 1);
        PP_TAIL(2, 1, 1, kt + 1);
        PP_TAIL(1, 2, 1, kt + 1);
        PP_TAIL(0, 3, 1, kt + 1);
        PP_TRC();
#undef PP_TAIL
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();           // (the barrier the second group took at the start)

    // ---- epilogue (gemm_epilogue.h): bias -> act -> + residual -> fp32 store and / or bf16 shadow
    const int64_t tile_off = (int64_t)z * g.strideC + (int64_t)(m0 + wr * 128) * g.ldc + (n0 + wc * 64);
    gemm_epilogue<4, 2, true>(acc, g.C ? g.C + tile_off : nullptr, g.C16 ? g.C16 + tile_off : nullptr,
                              g.residual ? g.residual + tile_off : nullptr, g.bias ? g.bias + (n0 + wc * 64) : nullptr, (int)g.ldc,
                              g.M - (m0 + wr * 128), g.N - (n0 + wc * 64), g.act, li, lh);
#ifdef W2V2_TUNING
    if (TRACE && trc && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        trc[trc_n++] = clock64();
        trc[31] = (unsigned long long)trc_n;
    }
#endif
#undef PP_TRC
}

template <int D, bool PRIO, bool TRACE = false>
int launch_pp(GemmPPArgs& g, int nbatch, hipStream_t s) {
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_pp_kernel<D, PRIO, TRACE>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS));
        attr_set = true;
    }
    dim3 grid(g.tiles_m * g.tiles_n, 1, nbatch);
    hipLaunchKernelGGL((gemm_bf16_pp_kernel<D, PRIO, TRACE>), grid, dim3(512), PP_LDS, s, g);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace

#ifdef W2V2_TUNING
extern unsigned long long* g_tune_trace;
#endif

// Shapes this kernel takes: both operands as aligned bf16 shadows, K a multiple of 128 with at least 4 K tiles (the loop is
// unrolled over K-tile pairs and peels the last pair), any M / N (edge tiles clamp their loads and guard their stores).
bool gemm_bf16_pp_ok(int M, int N, int K, int64_t lda, int64_t ldb16, int64_t strideA) {
    return M >= 1 && N >= 1 && K % 128 == 0 && K >= 256 && lda % 8 == 0 && ldb16 % 8 == 0 && strideA % 8 == 0 && 256 * lda < (1 << 29) &&
           256 * ldb16 < (1 << 29);
}

int launch_gemm_bf16_pp(const uint16_t* A16, int64_t lda, int64_t strideA, const uint16_t* B16, int64_t ldb16, float* C, uint16_t* C16,
                        int64_t ldc, int64_t strideC, const float* bias, const float* residual, int M, int N, int K, int nbatch, int act,
                        hipStream_t s) {
    W2V2_REQUIRE(A16 && B16 && (C || C16) && gemm_bf16_pp_ok(M, N, K, lda, ldb16, strideA), "gemm_bf16_pp: unsupported operands");
    GemmPPArgs g;
    g.A16 = A16; g.B16 = B16; g.C = C; g.C16 = C16; g.bias = bias; g.residual = residual;
    g.lda = lda; g.ldb16 = ldb16; g.ldc = ldc; g.strideA = strideA; g.strideC = strideC;
    g.M = M; g.N = N; g.K = K; g.act = act;
    g.tiles_m = (M + PP_BM - 1) / PP_BM;
    g.tiles_n = (N + PP_BN - 1) / PP_BN;
#ifdef W2V2_TUNING
    g.trace = g_tune_trace;
    if (g.trace) return launch_pp<5, true, true>(g, nbatch, s);
    switch (tune_int("W2V2_PP_D", 5) * 2 + (tune_int("W2V2_PP_PRIO", 1) ? 1 : 0)) {
        case 4: return launch_pp<2, false>(g, nbatch, s);
        case 5: return launch_pp<2, true>(g, nbatch, s);
        case 6: return launch_pp<3, false>(g, nbatch, s);
        case 7: return launch_pp<3, true>(g, nbatch, s);
        case 8: return launch_pp<4, false>(g, nbatch, s);
        case 9: return launch_pp<4, true>(g, nbatch, s);
        case 10: return launch_pp<5, false>(g, nbatch, s);
        case 12: return launch_pp<6, false>(g, nbatch, s);
        case 13: return launch_pp<6, true>(g, nbatch, s);
        default: break;
    }
#endif
    return launch_pp<5, true>(g, nbatch, s);
}

}  // namespace w2v2
