// fp32 GEMM on the bf16 matrix cores ("bf16x3", precision mode 2) with BOTH operands pre-split into three bf16 planes, on the
// software-pipelined 128 x 256 structure of gemm_bf16_sw.hip: four waves of 128 x 64 outputs per block, two blocks per CU, a
// ten-slot ring of 8-KiB LDS items filled by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no VALU) and retired by counted
// vmcnt waits, fragment reads threaded between the MFMAs.
//
// Arithmetic (gemm_split.hip has the error argument): x = x0 + x1 + x2 exactly, x0 = bf16(x), x1 = bf16(x - x0), x2 = x - x0 - x1;
// a product keeps the six terms of order <= 2.  Here the PRODUCER of an activation writes its three planes once (GEMM / LayerNorm /
// conv0 / attention epilogues), where gemm_split.hip loads fp32 rows through registers and re-splits them with 11 VALU operations
// per 4 elements once per column tile (9 x for q|k|v): 6 bytes per element streamed instead of 4 loaded + split.
//
// Stream.  A K tile is 32 deep: an image row is 64 B (four 16-byte slots, slot ^= (row >> 2) & 3: conflict-free ds_read_b128), an
// item is 128 rows = 8 KiB = one plane of the A tile or of half the B tile (columns of waves 0-1 | 2-3).  Nine items per K tile, in
// the order they are read:   A2 | B0a B0b | A1 | B1a B1b | A0 | B2a B2b     (Xp = plane p; a | b = column halves).
// Item s lives in ring slot s mod 10 and is requested as soon as item s - 10 has been read by every wave.
//
// Schedule of a wave per K tile: six TERMS of 16 MFMAs (both 16-deep k steps of the tile x 4 x 2 accumulators), each term one
// (plane of A, plane of B) pair, ordered so that an operand is read from LDS ONCE per K tile into a register slot that died one
// or more terms earlier -- two A slots (2 x 32 VGPRs) and two B slots (2 x 16 VGPRs) hold all six operands:
//     term        T0        T1        T2        T3        T4              T5
//     product     a2 b0     a1 b0     a1 b1     a0 b1     a0 b0           a0 b2
//     reads       a1        b1        a0        --        b2, a2'         b0'          (' = next K tile; a2' lands in the OTHER A
//     requests    2 items   1         2         1         0               3            slot than a2: the loop is unrolled by two tiles)
// A term = s_waitcnt vmcnt(N) (the items it reads have landed: own pieces) + lgkmcnt(0) (the previous term's reads: its operands) /
// s_barrier / 16 MFMAs with the reads and the LDS-DMA pieces between them.  tools/split_sw_schedule.py simulates the ring (landing,
// slot reuse, register liveness, the counted waits) for every K.
//
// Weights are stored as the kernel's LDS images (launch_split_weight_sw): [K / 32][plane][N][32] bf16 with the slot XOR applied, so
// a B piece is 1 KiB of consecutive memory; activation planes are row-major like the fp32 tensor (same lda / batch stride, in
// elements), `planeA` elements apart -- overlapping rows (the strided convolutions, lda < K) work as they do in fp32.
// Every output element sums its products in the same order whatever the tiling: results do not depend on M, N or the batch.
//
// FMT = PF_F16X2 (precision mode f16x2): the same ring with TWO fp16 planes per operand and THREE products per K step,
// a1 b0 + a0 b0 + a0 b1 -- half the MFMAs and two thirds of the operand bytes of bf16x3.  Why it exists: the matrix pipe of this chip is
// POWER-limited on real data (tools/mfma_power_probe.hip: a register-only bf16 MFMA loop sustains 0.68-0.76 of the nominal 2.5 PFLOP/s
// on random operands, the clock falls to 1.67-1.85 GHz), the six-product kernel above already keeps the pipe 0.79-0.86 busy and still
// only reaches 190-220 TFLOP/s fp32-equivalent sustained: the remaining lever is fewer MFMAs per fp32 product, not a better schedule.
// Six items per K tile:  A1 | B0a B0b | A0 | B1a B1b;  terms T0 a1 b0 (reads a0), T1 a0 b0 (reads b1, a1'), T2 a0 b1 (reads b0').
// Operands are scaled by powers of two before the split (activations by F16X2_ACT_SCALE, a weight by the exponent its image builder
// picked from max |w|: common.h, split_weight_sw); the epilogue multiplies the accumulators by the inverse (`out_scale`), exactly.
#include <utility>

#include "common.h"
#include "gemm_epilogue.h"
#include "gemm_sw_common.h"

namespace w2v2 {

namespace {

constexpr int SS_BM = 128, SS_BN = 256, SS_BK = 32;
constexpr int SS_ITEM = 8192, SS_SLOTS = 10, SS_LDS = SS_SLOTS * SS_ITEM;      // 80 KiB: two blocks per CU
constexpr int ss_ipt(int fmt) { return 3 * plane_count(fmt); }                   // items per K tile: 9 (bf16x3) | 6 (f16x2)
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

struct SplitSWArgs {
    const uint16_t* A16;       // plane 0 of A (M, K) rows lda apart; plane p at + p planeA
    const uint16_t* Bimg;      // the weight as LDS images (launch_split_weight_sw)
    float* C;
    uint16_t* C16;             // output planes (plane p at + p planeC), or null
    const float* bias;
    const float* residual;
    const float* out_scale;    // f16x2: device scalar the accumulators are multiplied by (1 / (activation scale x weight scale)); null = 1
    int* range_flag;           // f16x2 plane output: sticky flag set when a value saturates fp16 (may be null)
    int64_t planeA, planeC;
    int64_t lda, ldc, strideA, strideC;
    int M, N, K, act;
    int tiles_m, tiles_n;
    int gm = 0;                // grouped tile order: rows per group (common.h::grouped_tile); 0 = linear order
};

enum { SS_FULL = 0, SS_PENULT = 1, SS_LAST = 2 };

// acc = act(acc + bias) on all 128 values of a lane, every index a constant (a pragma-unrolled loop around 128 inlined erff bodies
// exceeds the unroller's size limit, and a loop that stays a loop indexes the accumulators dynamically: they would live in scratch)
template <int A, int I>
__device__ __forceinline__ void act_one(f32x16 (&acc)[4][2], float bv0, float bv1) {
    const float x = acc[(I >> 4) & 3][I >> 6][I & 15] + ((I >> 6) ? bv1 : bv0);
    acc[(I >> 4) & 3][I >> 6][I & 15] = A == 1 ? gelu_erf_select(x) : gelu_tanh_select(x);
}
template <int A, int... I>
__device__ __forceinline__ void act_all(f32x16 (&acc)[4][2], float bv0, float bv1, std::integer_sequence<int, I...>) {
    (act_one<A, I>(acc, bv0, bv1), ...);
}

// EK: 0 = epilogue from registers (ragged row tiles, odd strides), 1 + act = three output planes through LDS, 4 + act = fp32 output
// (+ residual) through LDS
template <int EK, int FMT>
__global__ __launch_bounds__(256, 2) void gemm_split_sw_kernel(SplitSWArgs g) {
    constexpr int NP = plane_count(FMT), IPT = ss_ipt(FMT);
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // the wave's 64-column group
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware tile order over the whole (batch, tile) space, the shorter grid dimension fastest (gemm_bf16_sw.hip)
    const int nwg = g.tiles_m * g.tiles_n;
    int bid, z;
    {
        const int total = nwg * (int)gridDim.z;
        int lin = (int)blockIdx.x + nwg * (int)blockIdx.z;
        const int q = total >> 3, r = total & 7, xcd = lin & 7, idx = lin >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        z = lin / nwg;
        bid = lin - z * nwg;
    }
    const bool m_fast = g.tiles_m < g.tiles_n;
    // (integer division runs on the VALU: readfirstlane brings the block-uniform tile coordinates back into SGPRs -- per-lane copies
    //  of them and of the offsets derived from them would have to be parked in scratch across the K loop)
    z = __builtin_amdgcn_readfirstlane(z);
    int tm_ = m_fast ? bid % g.tiles_m : bid / g.tiles_n, tn_ = m_fast ? bid / g.tiles_m : bid % g.tiles_n;
    if (g.gm > 0) grouped_tile(bid, g.tiles_m, g.tiles_n, g.gm, tm_, tn_);      // wide outputs: gm x (64 / gm) patches in flight (common.h)
    const int tm = __builtin_amdgcn_readfirstlane(tm_);
    const int tn = __builtin_amdgcn_readfirstlane(tn_);
    const int m0 = tm * SS_BM, n0 = tn * SS_BN;
    const int nk = g.K / SS_BK;                                     // even, >= 2

    // ---- LDS-DMA sources.  An item is 128 image rows = 8 pieces of 1 KiB (16 rows x 64 B), two per wave: piece i of wave w = image
    // rows 32 w + 16 i .. + 15; the lane at physical slot (lane & 3) of row r fetches logical slot (lane & 3) ^ ((r >> 2) & 3), and
    // (r >> 2) & 3 = (lane >> 4) & 3 for every piece.  A: per-lane byte offsets from a scalar base (rows clamped to the matrix: a
    // ragged last row tile re-reads row M - 1).  B: the images are stored swizzled, a piece is 1 KiB of consecutive memory.
    uint32_t offA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int ar = wave * 32 + i * 16 + (lane >> 2);
        ar = m0 + ar < g.M ? ar : g.M - 1 - m0;
        offA[i] = 2u * ((uint32_t)((int64_t)ar * g.lda) + (uint32_t)((((lane & 3) ^ (lane >> 4)) & 3) << 3));
    }
    const uint32_t offB0 = (uint32_t)lane * 16u, offB1 = offB0 + 1024u;
    auto uniform_ptr = [](const uint16_t* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const unsigned char*>(((uint64_t)hi << 32) | lo);
    };
    const unsigned char* const baseA = uniform_ptr(g.A16 + (int64_t)z * g.strideA + (int64_t)m0 * g.lda);
    const unsigned char* const baseB = uniform_ptr(g.Bimg + ((int64_t)n0 + 32 * wave) * SS_BK);
    const int64_t planeAb = 2 * g.planeA;                                  // bytes between planes of A
    const int64_t bplane = (int64_t)g.N * (2 * SS_BK), btile = NP * bplane;   // bytes between planes / K tiles of the weight images
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ss_smem;

    // piece I (0 | 1) of this wave's share of the stream item of kind J (= item number mod IPT) of K tile `ktile`, into ring slot `slot`.
    // Kinds: J % 3 == 0 is an A plane -- bf16x3: 0, 3, 6 = planes 2, 1, 0; f16x2: 0, 3 = planes 1, 0 -- else plane J / 3 of B, half J % 3 - 1.
    auto issue_piece = [&](auto Jc, auto Ic, int ktile, int slot) {
        constexpr int J = decltype(Jc)::value, I = decltype(Ic)::value;
        const unsigned dst = lds0 + (unsigned)slot * SS_ITEM + (unsigned)wave * 2048u + (unsigned)I * 1024u;
        if constexpr (J % 3 == 0) {
            constexpr int PL = NP - 1 - J / 3;
            sw_dma(dst, offA[I], baseA + PL * planeAb + (int64_t)ktile * (2 * SS_BK));
        } else {
            constexpr int PL = J / 3, HALF = (J % 3) - 1;
            sw_dma(dst, I == 0 ? offB0 : offB1, baseB + (int64_t)ktile * btile + PL * bplane + HALF * (128 * 2 * SS_BK));
        }
    };
    auto issue = [&](auto Jc, int ktile, int slot) {
        issue_piece(Jc, IC<0>{}, ktile, slot);
        issue_piece(Jc, IC<1>{}, ktile, slot);
    };

    // ---- fragment reads.  Image row rho = 32 rb + li (A) or 64 (wave & 1) + 32 jb + li (B): (rho >> 2) & 3 = (li >> 2) & 3 for
    // both.  Logical slot of k step ks, lane half lh = 2 ks + lh; physical = logical ^ s: address = slot base + x0 ^ (32 ks) + 2048 rb.
    const unsigned sz = (unsigned)(li >> 2) & 3u;
    const unsigned x0 = lds0 + (unsigned)li * 64u + ((((unsigned)lh ^ sz) & 1u) << 4) + ((sz >> 1) << 5);
    const unsigned bwave = (unsigned)(wave & 1) * 4096u;          // this wave's 64 rows inside its B item
    const int half = wave >> 1;                                   // this wave's B item of a pair: a (waves 0-1) | b (waves 2-3)

    bf16x8 fa[2][4][2];      // [A register slot][32-row block][k step]
    bf16x8 fb[2][2][2];      // [B register slot][32-column block][k step]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 zero = {};
            fa[s][0][ks] = fa[s][1][ks] = fa[s][2][ks] = fa[s][3][ks] = fb[s][0][ks] = fb[s][1][ks] = zero;
        }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    auto slot_of = [](int s0, int j) {      // (s0 + j) mod 10 for j < 20
        int s = s0 + j;
        s = s >= SS_SLOTS ? s - SS_SLOTS : s;
        return s >= SS_SLOTS ? s - SS_SLOTS : s;
    };
    // read unit U (0 .. 7) of an A item into A register slot S: k step U >> 2, 32-row block U & 3
    auto rd_a = [&](auto Sc, auto Uc, int slot) {
        constexpr int S = decltype(Sc)::value, U = decltype(Uc)::value, KS = U >> 2, RB = U & 3;
        fa[S][RB][KS] = sw_read<RB * 2048>((x0 ^ (32u * KS)) + (unsigned)slot * SS_ITEM);
    };
    // read unit U (0 .. 3) of this wave's B item into B register slot S: k step U >> 1, 32-column block U & 1
    auto rd_b = [&](auto Sc, auto Uc, int slot) {
        constexpr int S = decltype(Sc)::value, U = decltype(Uc)::value, KS = U >> 1, JB = U & 1;
        fb[S][JB][KS] = sw_read<JB * 2048>((x0 ^ (32u * KS)) + (unsigned)slot * SS_ITEM + bwave);
    };
    // MFMA m (0 .. 15) of a term on A slot SA, B slot SB: k step m >> 3, row block (m >> 1) & 3, column block m & 1
    auto mm = [&](auto SAc, auto SBc, auto Mc) {
        constexpr int SA = decltype(SAc)::value, SB = decltype(SBc)::value, M = decltype(Mc)::value;
        constexpr int KS = M >> 3, RB = (M >> 1) & 3, JB = M & 1;
        if constexpr (FMT == PF_F16X2)
            acc[RB][JB] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[SA][RB][KS]), __builtin_bit_cast(f16x8, fb[SB][JB][KS]), acc[RB][JB], 0, 0, 0);
        else
            acc[RB][JB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SA][RB][KS], fb[SB][JB][KS], acc[RB][JB], 0, 0, 0);
    };
    // wait for every fragment read in flight and pin the fragment registers behind the wait (the MFMAs cannot move above it)
#define SS_TIE_ALL()                                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                               \
                 : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]), "+v"(fa[0][2][0]), "+v"(fa[0][2][1]),          \
                   "+v"(fa[0][3][0]), "+v"(fa[0][3][1]), "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[1][1][0]), "+v"(fa[1][1][1]),          \
                   "+v"(fa[1][2][0]), "+v"(fa[1][2][1]), "+v"(fa[1][3][0]), "+v"(fa[1][3][1]), "+v"(fb[0][0][0]), "+v"(fb[0][0][1]),          \
                   "+v"(fb[0][1][0]), "+v"(fb[0][1][1]), "+v"(fb[1][0][0]), "+v"(fb[1][0][1]), "+v"(fb[1][1][0]), "+v"(fb[1][1][1])           \
                 :: "memory")

    // A term: s_waitcnt vmcnt(VM) (the items it reads have landed: this wave's pieces) + lgkmcnt(0) (the previous term's reads = its
    // operands; also: their ring slots are free) / s_barrier / 16 MFMAs on A slot SA x B slot SB with work unit U issued behind MFMA U.
    auto term_body = [&](auto VMc, auto SAc, auto SBc, auto&& unit_u) {
        sw_wait_vm<decltype(VMc)::value>();
        SS_TIE_ALL();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#define SS_STEP(U)                                                    \
        mm(SAc, SBc, IC<U>{});                                        \
        unit_u(IC<U>{});                                              \
        __builtin_amdgcn_sched_barrier(0);
        SS_STEP(0) SS_STEP(1) SS_STEP(2) SS_STEP(3) SS_STEP(4) SS_STEP(5) SS_STEP(6) SS_STEP(7)
        SS_STEP(8) SS_STEP(9) SS_STEP(10) SS_STEP(11) SS_STEP(12) SS_STEP(13) SS_STEP(14) SS_STEP(15)
#undef SS_STEP
    };
    int s0 = 0, kt = 0;
    auto next_s0 = [](int s) { return s + IPT >= SS_SLOTS ? s + IPT - SS_SLOTS : s + IPT; };

    if constexpr (FMT == PF_BF16X3) {
    // Work unit U of term T: the term's reads first, then its LDS-DMA pieces.
    //   T0: a1 (item 3) -> A slot PAR^1 | items 11, 12          T1: b1 (item 4|5) -> B slot 1 | item 13
    //   T2: a0 (item 6) -> A slot PAR   | items 14, 15          T3: -- | item 16
    //   T4: b2 (item 7|8) -> B slot 1, a2' (item 9) -> A slot PAR^1 | --          T5: b0' (item 10|11) -> B slot 0 | items 17, 18, 19
    // Item 9 kappa + j is of kind j mod 9 and belongs to K tile kappa + j / 9.  MODE: the last two K tiles request / read only what exists.
    auto unit = [&](auto Tc, auto PARc, auto MODEc, auto Uc, int s0, int kt) {
        constexpr int T = decltype(Tc)::value, PAR = decltype(PARc)::value, MODE = decltype(MODEc)::value, U = decltype(Uc)::value;
        constexpr bool REQ = MODE != SS_LAST;          // requests exist (the penultimate tile: all but items 18, 19)
        if constexpr (T == 0) {
            if constexpr (U < 8) rd_a(IC<PAR ^ 1>{}, IC<U>{}, slot_of(s0, 3));
            else if constexpr (U < 10 && REQ) issue_piece(IC<2>{}, IC<U - 8>{}, kt + 1, slot_of(s0, 11));
            else if constexpr (U < 12 && REQ) issue_piece(IC<3>{}, IC<U - 10>{}, kt + 1, slot_of(s0, 12));
        } else if constexpr (T == 1) {
            if constexpr (U < 4) rd_b(IC<1>{}, IC<U>{}, slot_of(s0, 4 + half));
            else if constexpr (U < 6 && REQ) issue_piece(IC<4>{}, IC<U - 4>{}, kt + 1, slot_of(s0, 13));
        } else if constexpr (T == 2) {
            if constexpr (U < 8) rd_a(IC<PAR>{}, IC<U>{}, slot_of(s0, 6));
            else if constexpr (U < 10 && REQ) issue_piece(IC<5>{}, IC<U - 8>{}, kt + 1, slot_of(s0, 14));
            else if constexpr (U < 12 && REQ) issue_piece(IC<6>{}, IC<U - 10>{}, kt + 1, slot_of(s0, 15));
        } else if constexpr (T == 3) {
            if constexpr (U < 2 && REQ) issue_piece(IC<7>{}, IC<U>{}, kt + 1, slot_of(s0, 16));
        } else if constexpr (T == 4) {
            if constexpr (U < 4) rd_b(IC<1>{}, IC<U>{}, slot_of(s0, 7 + half));
            else if constexpr (U < 12 && MODE != SS_LAST) rd_a(IC<PAR ^ 1>{}, IC<U - 4>{}, slot_of(s0, 9));
        } else {
            if constexpr (U < 4 && MODE != SS_LAST) rd_b(IC<0>{}, IC<U>{}, slot_of(s0, 10 + half));
            else if constexpr (U >= 4 && U < 6 && REQ) issue_piece(IC<8>{}, IC<U - 4>{}, kt + 1, slot_of(s0, 17));
            else if constexpr (U >= 6 && U < 8 && MODE == SS_FULL) issue_piece(IC<0>{}, IC<U - 6>{}, kt + 2, slot_of(s0, 18));
            else if constexpr (U >= 8 && U < 10 && MODE == SS_FULL) issue_piece(IC<1>{}, IC<U - 8>{}, kt + 2, slot_of(s0, 19));
        }
    };
    // The counted wait: the ring holds 10 - r(T-1) unread items when term T begins (r = items a term reads: 1 2 1 0 3 2), the oldest
    // r(T) of them must have landed -> 2 (10 - r(T-1) - r(T)) of this wave's pieces may stay in flight; in the last K tile nothing
    // younger is requested any more and the counts run down to zero.
    auto term = [&](auto Tc, auto PARc, auto MODEc, int s0, int kt) {
        constexpr int T = decltype(Tc)::value, PAR = decltype(PARc)::value, MODE = decltype(MODEc)::value;
        constexpr int VM = MODE != SS_LAST ? (T == 3 ? -1 : T == 5 ? 10 : 14) : (T == 0 ? 10 : T == 1 ? 6 : T == 2 ? 4 : T == 4 ? 0 : -1);
        constexpr int SA = (T == 0 || T >= 3) ? PAR : PAR ^ 1;      // a2, a0 live in A slot PAR, a1 in the other
        constexpr int SB = (T == 2 || T == 3 || T == 5) ? 1 : 0;   // b0 in B slot 0; b1, then b2, in slot 1
        term_body(IC<VM>{}, IC<SA>{}, IC<SB>{}, [&](auto Uc) { unit(Tc, PARc, MODEc, Uc, s0, kt); });
    };
    auto tile = [&](auto PARc, auto MODEc, int s0, int kt) {
        term(IC<0>{}, PARc, MODEc, s0, kt);
        term(IC<1>{}, PARc, MODEc, s0, kt);
        term(IC<2>{}, PARc, MODEc, s0, kt);
        term(IC<3>{}, PARc, MODEc, s0, kt);
        term(IC<4>{}, PARc, MODEc, s0, kt);
        term(IC<5>{}, PARc, MODEc, s0, kt);
    };

    // ---- prologue: ten items in flight (K tile 0 and a2 of K tile 1); then the two read-only "terms" in front of K tile 0
    issue(IC<0>{}, 0, 0); issue(IC<1>{}, 0, 1); issue(IC<2>{}, 0, 2); issue(IC<3>{}, 0, 3); issue(IC<4>{}, 0, 4);
    issue(IC<5>{}, 0, 5); issue(IC<6>{}, 0, 6); issue(IC<7>{}, 0, 7); issue(IC<8>{}, 0, 8); issue(IC<0>{}, 1, 9);
    sw_wait_vm<18>();                                                 // item 0 (nine younger items of two pieces each may be in flight)
    __builtin_amdgcn_s_barrier();
    rd_a(IC<0>{}, IC<0>{}, 0); rd_a(IC<0>{}, IC<1>{}, 0); rd_a(IC<0>{}, IC<2>{}, 0); rd_a(IC<0>{}, IC<3>{}, 0);
    rd_a(IC<0>{}, IC<4>{}, 0); rd_a(IC<0>{}, IC<5>{}, 0); rd_a(IC<0>{}, IC<6>{}, 0); rd_a(IC<0>{}, IC<7>{}, 0);
    sw_wait_vm<14>();                                                 // items 1, 2
    SS_TIE_ALL();
    __builtin_amdgcn_s_barrier();
    issue(IC<1>{}, 1, 0);                                             // item 10 (slot 0: every wave's a2 reads retired before the barrier)
    rd_b(IC<0>{}, IC<0>{}, 1 + half); rd_b(IC<0>{}, IC<1>{}, 1 + half); rd_b(IC<0>{}, IC<2>{}, 1 + half); rd_b(IC<0>{}, IC<3>{}, 1 + half);

    __builtin_amdgcn_s_setprio(1);                                    // the K loop outranks the co-resident block's epilogue on this SIMD
    for (; kt + 2 < nk; kt += 2) {
        tile(IC<0>{}, IC<SS_FULL>{}, s0, kt);
        s0 = next_s0(s0);
        tile(IC<1>{}, IC<SS_FULL>{}, s0, kt + 1);
        s0 = next_s0(s0);
    }
    tile(IC<0>{}, IC<SS_PENULT>{}, s0, kt);
    s0 = next_s0(s0);
    tile(IC<1>{}, IC<SS_LAST>{}, s0, kt + 1);
    __builtin_amdgcn_s_setprio(0);
    } else {
    // ---- f16x2: six items per K tile (kinds 0 A1 | 1, 2 B0a, B0b | 3 A0 | 4, 5 B1a, B1b), three terms.  a1 lives in A slot 0, a0 in
    // slot 1, b0 in B slot 0, b1 in slot 1 -- no parity.  Item 6 kappa + j is of kind j mod 6 and belongs to K tile kappa + j / 6.
    //   T0 a1 b0: reads a0 (item 3) -> A slot 1 | requests items 11, 12
    //   T1 a0 b0: reads b1 (item 4|5) -> B slot 1, a1' (item 6) -> A slot 0 | item 13
    //   T2 a0 b1: reads b0' (item 7|8) -> B slot 0 | items 14, 15, 16
    // Items read per term r = 1 3 2: counted waits 2 (10 - r(T-1) - r(T)) = 14 12 10; the penultimate tile requests only item 11 (the last
    // tile's B1b), the last one nothing: 14 10 6, then 4 0 -.
    auto unit2 = [&](auto Tc, auto MODEc, auto Uc, int s0, int kt) {
        constexpr int T = decltype(Tc)::value, MODE = decltype(MODEc)::value, U = decltype(Uc)::value;
        if constexpr (T == 0) {
            if constexpr (U < 8) rd_a(IC<1>{}, IC<U>{}, slot_of(s0, 3));
            else if constexpr (U < 10 && MODE != SS_LAST) issue_piece(IC<5>{}, IC<U - 8>{}, kt + 1, slot_of(s0, 11));
            else if constexpr (U >= 10 && U < 12 && MODE == SS_FULL) issue_piece(IC<0>{}, IC<U - 10>{}, kt + 2, slot_of(s0, 12));
        } else if constexpr (T == 1) {
            if constexpr (U < 4) rd_b(IC<1>{}, IC<U>{}, slot_of(s0, 4 + half));
            else if constexpr (U < 12 && MODE != SS_LAST) rd_a(IC<0>{}, IC<U - 4>{}, slot_of(s0, 6));
            else if constexpr (U >= 12 && U < 14 && MODE == SS_FULL) issue_piece(IC<1>{}, IC<U - 12>{}, kt + 2, slot_of(s0, 13));
        } else {
            if constexpr (U < 4 && MODE != SS_LAST) rd_b(IC<0>{}, IC<U>{}, slot_of(s0, 7 + half));
            else if constexpr (U >= 4 && U < 6 && MODE == SS_FULL) issue_piece(IC<2>{}, IC<U - 4>{}, kt + 2, slot_of(s0, 14));
            else if constexpr (U >= 6 && U < 8 && MODE == SS_FULL) issue_piece(IC<3>{}, IC<U - 6>{}, kt + 2, slot_of(s0, 15));
            else if constexpr (U >= 8 && U < 10 && MODE == SS_FULL) issue_piece(IC<4>{}, IC<U - 8>{}, kt + 2, slot_of(s0, 16));
        }
    };
    auto term2 = [&](auto Tc, auto MODEc, int s0, int kt) {
        constexpr int T = decltype(Tc)::value, MODE = decltype(MODEc)::value;
        constexpr int VM = MODE == SS_FULL ? (T == 0 ? 14 : T == 1 ? 12 : 10) : MODE == SS_PENULT ? (T == 0 ? 14 : T == 1 ? 10 : 6) : (T == 0 ? 4 : T == 1 ? 0 : -1);
        constexpr int SA = T == 0 ? 0 : 1, SB = T == 2 ? 1 : 0;
        term_body(IC<VM>{}, IC<SA>{}, IC<SB>{}, [&](auto Uc) { unit2(Tc, MODEc, Uc, s0, kt); });
    };
    auto tile2 = [&](auto MODEc, int s0, int kt) {
        term2(IC<0>{}, MODEc, s0, kt);
        term2(IC<1>{}, MODEc, s0, kt);
        term2(IC<2>{}, MODEc, s0, kt);
    };
    // prologue: K tile 0 and items 0 .. 3 of K tile 1 in flight; a1, then b0, of K tile 0 read
    issue(IC<0>{}, 0, 0); issue(IC<1>{}, 0, 1); issue(IC<2>{}, 0, 2); issue(IC<3>{}, 0, 3); issue(IC<4>{}, 0, 4);
    issue(IC<5>{}, 0, 5); issue(IC<0>{}, 1, 6); issue(IC<1>{}, 1, 7); issue(IC<2>{}, 1, 8); issue(IC<3>{}, 1, 9);
    sw_wait_vm<18>();
    __builtin_amdgcn_s_barrier();
    rd_a(IC<0>{}, IC<0>{}, 0); rd_a(IC<0>{}, IC<1>{}, 0); rd_a(IC<0>{}, IC<2>{}, 0); rd_a(IC<0>{}, IC<3>{}, 0);
    rd_a(IC<0>{}, IC<4>{}, 0); rd_a(IC<0>{}, IC<5>{}, 0); rd_a(IC<0>{}, IC<6>{}, 0); rd_a(IC<0>{}, IC<7>{}, 0);
    sw_wait_vm<14>();
    SS_TIE_ALL();
    __builtin_amdgcn_s_barrier();
    issue(IC<4>{}, 1, 0);                                             // item 10 = B1a of K tile 1
    rd_b(IC<0>{}, IC<0>{}, 1 + half); rd_b(IC<0>{}, IC<1>{}, 1 + half); rd_b(IC<0>{}, IC<2>{}, 1 + half); rd_b(IC<0>{}, IC<3>{}, 1 + half);

    __builtin_amdgcn_s_setprio(1);
    for (; kt + 2 < nk; ++kt) {
        tile2(IC<SS_FULL>{}, s0, kt);
        s0 = next_s0(s0);
    }
    tile2(IC<SS_PENULT>{}, s0, kt);
    s0 = next_s0(s0);
    tile2(IC<SS_LAST>{}, s0, kt + 1);
    __builtin_amdgcn_s_setprio(0);
    }
    // (every wave's last LDS reads retired before the barrier of the last term and every piece has landed: the ring is free)

    // ---- epilogue: bias -> act (erff) -> + residual -> fp32 store, or the planes of the result for the next GEMM
    // (the lane index is laundered so that no per-lane epilogue address is computed -- and parked in scratch -- in front of the K loop)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int li_e = lane_e & 31, lh_e = lane_e >> 5;
    const int64_t tile_off = (int64_t)z * g.strideC + (int64_t)m0 * g.ldc + (n0 + wave * 64);
    const bool whole = g.M - m0 >= 128;                               // (block-uniform; ragged last row tiles take the register epilogue)
    const float* const bw = g.bias ? g.bias + (n0 + wave * 64) : nullptr;
    const unsigned wb = lds0 + (unsigned)wave * 16384u;
    if constexpr (FMT == PF_F16X2) {      // undo the operands' power-of-two scales (exact)
        const float sc = g.out_scale ? *g.out_scale : 1.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= sc;
    }
    // the activation is applied in place first (one copy of erff per instance; the stores below then see act = 0)
    constexpr int ACT = EK == 0 ? -1 : (EK - 1) % 3;
    const float* bw2 = bw;
    auto activate = [&](auto ACTc) {
        act_all<decltype(ACTc)::value>(acc, bw ? bw[li_e] : 0.0f, bw ? bw[32 + li_e] : 0.0f, std::make_integer_sequence<int, 128>{});
        bw2 = nullptr;
    };
    if constexpr (ACT == 1 || ACT == 2) activate(IC<ACT>{});      // (EK 0 carries no activation: the launcher refuses act != 0 on unaligned outputs)
    auto from_registers = [&]() {
        // (what an instance cannot have is null at compile time: the plane instances have no fp32 output / residual, the fp32 ones no planes)
        float* const c32 = (EK >= 1 && EK <= 3) ? nullptr : (g.C ? g.C + tile_off : nullptr);
        const float* const r32 = (EK >= 1 && EK <= 3) ? nullptr : (g.residual ? g.residual + tile_off : nullptr);
        uint16_t* const c16 = EK >= 4 ? nullptr : (g.C16 ? g.C16 + tile_off : nullptr);
        gemm_epilogue<4, 2, false>(acc, c32, c16, r32, bw2, (int)g.ldc, g.M - m0, g.N - (n0 + wave * 64), 0, li_e, lh_e, c16 ? g.planeC : 0, FMT, g.range_flag);
    };
    if constexpr (EK >= 1 && EK <= 3) {
        if (whole) sw_epilogue_planes<FMT>(false, acc, g.C16 + tile_off, g.planeC, bw2, (int)g.ldc, wb, lane_e, g.range_flag);
        else from_registers();
    } else if constexpr (EK >= 4) {
        if (whole) sw_epilogue_f32<0, false>(false, acc, g.C + tile_off, nullptr, g.residual ? g.residual + tile_off : nullptr, bw2, (int)g.ldc, wb, lane_e);
        else from_registers();
    } else {
        from_registers();
    }
#undef SS_TIE_ALL
}

// w (K, N) row-major fp32  ->  the kernel's LDS images: [K / 32][plane][N][32] 16-bit terms, the 16-byte slot of row n XOR-ed with
// (n >> 2) & 3.  FMT PF_BF16X3: three bf16 planes, exact.  PF_F16X2: two fp16 planes of w 2^e, e = 14 - exponent(max |w|) read from
// `wmax_bits` (the bit pattern of max |w|, written by absmax_kernel in front); block (0, 0) leaves the matching accumulator scale
// 1 / (2^e x F16X2_ACT_SCALE) in *out_scale for the GEMM's epilogue.
template <int FMT>
__global__ __launch_bounds__(256) void split_weight_sw_kernel(const float* __restrict__ w, uint16_t* __restrict__ img, int K, int N,
                                                              const unsigned* __restrict__ wmax_bits, float* __restrict__ out_scale) {
    __shared__ float tile[64][65];
    constexpr int NP = plane_count(FMT);
    const int k0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    float ws = 1.0f;
    if constexpr (FMT == PF_F16X2) {
        const unsigned mb = *wmax_bits;                                      // max |w| as fp32 bits
        int e = 14 + 127 - (int)((mb >> 23) & 0xffu);                        // max |w| 2^e in [2^14, 2^15): below fp16's 65504
        e = mb == 0u ? 0 : (e > 100 ? 100 : (e < -100 ? -100 : e));
        ws = __uint_as_float((unsigned)(127 + e) << 23);
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *out_scale = __uint_as_float((unsigned)(127 - e) << 23) / F16X2_ACT_SCALE;
    }
    for (int r = ty; r < 64; r += 4) {
        const int k = k0 + r, n = n0 + tx;
        tile[r][tx] = (k < K && n < N) ? w[(int64_t)k * N + n] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int n = n0 + r, k = k0 + tx;
        if (n < N && k < K) {
            const float x = tile[tx][r];
            const int kt = k / SS_BK, kk = k % SS_BK;
            const int64_t plane = (int64_t)N * SS_BK;
            const int64_t o = ((int64_t)kt * NP * N + n) * SS_BK + (((kk >> 3) ^ ((n >> 2) & 3)) << 3) + (kk & 7);
            if constexpr (FMT == PF_F16X2) {
                bool ovf = false;
                split2h_one(x, ws, img[o], img[o + plane], ovf);             // (cannot saturate: |x| ws < 2^15)
            } else {
                split3_one(x, img[o], img[o + plane], img[o + 2 * plane]);
            }
        }
    }
}

// max |w| over n elements as fp32 bits (non-negative floats order like their bit patterns); *out zeroed by the launcher
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ w, int64_t n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float a = fabsf(w[i]);
        m = a > m ? a : m;                                                    // (NaN never wins; inf does, and clamps the exponent)
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// x (n fp32, 16-byte aligned, n % 4 == 0) -> its planes (tests, and producers that have no fused form)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, uint16_t* __restrict__ p, int64_t plane, int64_t n4, int fmt,
                                                           int* range_flag) {
    bool ovf = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        store_planes4(p + 4 * i, plane, fmt, reinterpret_cast<const f32x4_t*>(x)[i], ovf);
    report_overflow(range_flag, ovf);
}

template <int EK, int FMT>
int launch_ss(SplitSWArgs& g, dim3 grid, hipStream_t s) {
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_sw_kernel<EK, FMT>), hipFuncAttributeMaxDynamicSharedMemorySize, SS_LDS));
        attr_set = true;
    }
    W2V2_LAUNCH((gemm_split_sw_kernel<EK, FMT>), grid, dim3(256), SS_LDS, s, g);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}
template <int FMT>
int launch_ss_ek(int ek, SplitSWArgs& g, dim3 grid, hipStream_t s) {
    switch (ek) {
        case 1: return launch_ss<1, FMT>(g, grid, s);
        case 2: return launch_ss<2, FMT>(g, grid, s);
        case 3: return launch_ss<3, FMT>(g, grid, s);
        case 4: return launch_ss<4, FMT>(g, grid, s);
        case 5: return launch_ss<5, FMT>(g, grid, s);
        case 6: return launch_ss<6, FMT>(g, grid, s);
        default: return launch_ss<0, FMT>(g, grid, s);
    }
}

}  // namespace

// Shapes: whole 256-column tiles, K a multiple of 64 (the K loop runs in pairs of 32-deep tiles), 16-byte aligned plane rows, any M.
bool gemm_split_sw_ok(const uint16_t* A16, int64_t planeA, int64_t lda, int64_t strideA, int M, int N, int K) {
    return A16 && M >= 1 && N >= 256 && N % 256 == 0 && K >= 64 && K % 64 == 0 && lda % 8 == 0 && strideA % 8 == 0 && planeA % 8 == 0 &&
           (reinterpret_cast<uintptr_t>(A16) & 15) == 0 && 128 * lda < (1 << 29);
}

// `scratch`: f16x2 only -- two device words owned by the caller next to the images: [0] max |w| bits (work), [1] the accumulator scale (fp32)
int launch_split_weight_sw(const float* w, uint16_t* img, int K, int N, int fmt, void* scratch, hipStream_t s) {
    W2V2_REQUIRE(w && img && K > 0 && N > 0, "split_weight_sw: bad argument");
    W2V2_REQUIRE(K % 64 == 0 && N % SS_BN == 0, "split_weight_sw: needs K %% 64 == 0 and N %% 256 == 0");
    const dim3 grid((N + 63) / 64, (K + 63) / 64);
    if (fmt == PF_F16X2) {
        W2V2_REQUIRE(scratch, "split_weight_sw: the f16x2 format needs its two scratch words");
        unsigned* const bits = reinterpret_cast<unsigned*>(scratch);
        W2V2_HIP_CHECK(hipMemsetAsync(bits, 0, sizeof(unsigned), s));
        const int64_t n = (int64_t)K * N;
        W2V2_LAUNCH(absmax_kernel, dim3((unsigned)((n + 256 * 16 - 1) / (256 * 16) > 1024 ? 1024 : (n + 256 * 16 - 1) / (256 * 16))), dim3(256), 0, s, w, n, bits);
        W2V2_LAUNCH(split_weight_sw_kernel<PF_F16X2>, grid, dim3(256), 0, s, w, img, K, N, bits, reinterpret_cast<float*>(bits + 1));
    } else {
        W2V2_LAUNCH(split_weight_sw_kernel<PF_BF16X3>, grid, dim3(256), 0, s, w, img, K, N, nullptr, nullptr);
    }
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_split_planes(const float* x, uint16_t* planes, int64_t plane, int64_t n, int fmt, int* range_flag, hipStream_t s) {
    W2V2_REQUIRE(x && planes && n > 0 && n % 4 == 0 && plane >= n && plane % 4 == 0, "split_planes: bad argument");
    W2V2_REQUIRE(((reinterpret_cast<uintptr_t>(x) & 15) | (reinterpret_cast<uintptr_t>(planes) & 7)) == 0, "split_planes: unaligned buffers");
    const int64_t n4 = n / 4;
    const int blocks = (n4 + 255) / 256 > 4096 ? 4096 : (int)((n4 + 255) / 256);
    ProfScope ps(tl_step_prof, FAM_MISC, 0.0, (4.0 + 2.0 * plane_count(fmt)) * (double)n, s);
    W2V2_LAUNCH(split_planes_kernel, dim3(blocks), dim3(256), 0, s, x, planes, plane, n4, fmt, range_flag);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_gemm_split_sw(Profiler* prof, int fmt, const uint16_t* A16, int64_t planeA, int64_t lda, int64_t strideA, const uint16_t* Bimg,
                         const float* out_scale, float* C, uint16_t* C16, int64_t planeC, int64_t ldc, int64_t strideC, const float* bias,
                         const float* residual, int M, int N, int K, int nbatch, int act, int* range_flag, hipStream_t s) {
    W2V2_REQUIRE(Bimg && (C || C16) && !(C && C16) && nbatch > 0, "gemm_split_sw: null operand (one of C / C16)");
    W2V2_REQUIRE(fmt == PF_BF16X3 || (fmt == PF_F16X2 && out_scale), "gemm_split_sw: unknown plane format / f16x2 without its scale");
    W2V2_REQUIRE(gemm_split_sw_ok(A16, planeA, lda, strideA, M, N, K), "gemm_split_sw: needs N %% 256 == 0, K %% 64 == 0, 16-byte aligned plane rows");
    W2V2_REQUIRE(ldc >= N && ldc < (1 << 23) && act >= 0 && act <= 2, "gemm_split_sw: bad leading dimension / activation");
    W2V2_REQUIRE((reinterpret_cast<uintptr_t>(Bimg) & 15) == 0 && !(C16 && residual), "gemm_split_sw: unaligned weight images / residual with plane output");
    SplitSWArgs g;
    g.A16 = A16; g.Bimg = Bimg; g.C = C; g.C16 = C16; g.bias = bias; g.residual = residual;
    g.out_scale = fmt == PF_F16X2 ? out_scale : nullptr; g.range_flag = range_flag;
    g.planeA = planeA; g.planeC = planeC;
    g.lda = lda; g.ldc = ldc; g.strideA = strideA; g.strideC = strideC;
    g.M = M; g.N = N; g.K = K; g.act = act;
    g.tiles_m = (M + SS_BM - 1) / SS_BM;
    g.tiles_n = N / SS_BN;
    g.gm = nbatch == 1 ? tile_group_rows(g.tiles_m, g.tiles_n, (int64_t)SS_BM * K * 2 * plane_count(fmt), 64) : 0;      // (A row panel: all its planes)
    const double np = plane_count(fmt);
    ProfScope ps(prof, FAM_GEMM_SPLIT, 2.0 * M * (double)N * K * nbatch,
                 nbatch * ((double)M * K * 2.0 * np + (double)M * N * (C ? 4.0 : 2.0 * np)) + 2.0 * np * (double)K * N, s);
    auto al = [](const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    const bool ldsp = C16 && ldc % 8 == 0 && strideC % 8 == 0 && planeC % 8 == 0 && al(C16, 16);
    const bool lds32 = C && ldc % 4 == 0 && strideC % 4 == 0 && al(C, 16) && (!residual || al(residual, 16));
    const int ek = ldsp ? 1 + act : lds32 ? 4 + act : 0;
    W2V2_REQUIRE(ek != 0 || act == 0, "gemm_split_sw: an activation needs 16-byte aligned output rows (ldc, batch stride, base pointers)");
    dim3 grid(g.tiles_m * g.tiles_n, 1, nbatch);
    return fmt == PF_F16X2 ? launch_ss_ek<PF_F16X2>(ek, g, grid, s) : launch_ss_ek<PF_BF16X3>(ek, g, grid, s);
}

}  // namespace w2v2
