// bf16 GEMM, 128 x 256 x 64 tiles, FOUR waves per block and TWO blocks per CU, every wave software-pipelined: the LDS reads of
// the next quadrant's operands and the LDS-DMA of the stream ten 8-KiB items ahead are issued between the MFMAs of the current
// quadrant.
//
// Why (profiles/r03_gemm_bf16_study.md): the 256 x 256 ping-pong kernel (gemm_bf16_pp.hip) runs its K loop at ~1.7 PF, but with
// K = 512 ... 1536 a tile is only 8 ... 24 K tiles long and its prologue (3 k cycles) and epilogue (10 k cycles for a bf16-only
// store, 16 k with GELU: ~20 VALU ops per output) run with the matrix pipe idle -- one block per CU, both waves of a SIMD in the
// same tile.  Two independent blocks per CU put the epilogue of one under the K loop of the other, which needs (a) half the LDS
// and registers per block: 4 waves of 128 x 64 outputs (128 accumulator VGPRs, 256 per wave at 2 waves / SIMD) and an 80-KiB
// operand ring, and (b) a wave that keeps the matrix pipe busy ON ITS OWN while its neighbour is in an epilogue -- no load segment
// that waits for a partner's MFMAs, but reads and DMA issue threaded through its own MFMA stream.
//
// Stream: per K tile six items of 8 KiB (64 image rows x 128 B) in the order they are read: A_0 | B_0a B_0b | B_1a B_1b | A_1
// (A_i = rows 64 i .. 64 i + 63 of the tile; B_j = columns 64 w + 32 j .. + 31 of every wave w, halves a / b = waves 0-1 / 2-3).
// Item s lives in ring slot s mod 10; it is requested as soon as item s - 10 has been read.  Quadrant order per K tile
// (0,0) (0,1) (1,1) (1,0): each quadrant's MFMAs free exactly the registers the reads issued beside them refill (B_1 | A_1 |
// next A_0 | next B_0, the last one k-step by k-step behind the MFMAs that consume the old values).
// A phase = s_waitcnt vmcnt(N_q) + lgkmcnt(0) / s_barrier (4 waves) / DMA issue / 8 MFMAs with 4 or 8 ds_read_b128 between them.
//
// Same arithmetic as gemm_bf16_kernel (k ascending in 16-deep MFMA steps, same lane -> k assignment): identical bits.
#include "common.h"
#include "gemm_epilogue.h"
#include "gemm_sw_common.h"

namespace w2v2 {

namespace {

constexpr int SW_BM = 128, SW_BN = 256, SW_BK = 64;
constexpr int SW_ITEM = 8192, SW_SLOTS = 10, SW_LDS = SW_SLOTS * SW_ITEM;      // 80 KiB: two blocks per CU

struct GemmSWArgs {
    const uint16_t* A16;
    const uint16_t* B16;       // (N, K) rows ldb16 apart; transposed form: (K, N) rows ldb16 apart, batch stride strideB
    float* C;
    uint16_t* C16;
    const float* bias;
    const float* residual;
    int64_t lda, ldb16, ldc, strideA, strideC, strideB;
    int kextra = 0;            // transposed form: batches z < kextra run one K tile more; batch z starts min(z, kextra) K tiles after z K rows
    int krag = 0;              // transposed form: rows that exist in the LAST K tile of the LAST batch (0 = all 64).  Rows past them are read
                               // from A's last existing row (finite) and from B's row just past the end, which the caller keeps all-zero
    int M, N, K, act;
    int tiles_m, tiles_n;
    int gm = 0;                // grouped tile order: rows per group (common.h::grouped_tile); 0 = linear order
#ifdef W2V2_TUNING
    unsigned long long* trace;
    int abl;
    int nt;                    // W2V2_SW_NT: see SW_NT_DEFAULT
#endif
};

__device__ __forceinline__ int sw_swz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 1); }      // = gemm_bf16.hip's swz

// EK: the epilogue this instance carries (one per instance: with all of them inlined into one kernel their pointers and strides stay
// live across the K loop and the allocator starts parking registers in scratch): 0 = from registers (ragged row tiles, odd strides),
// 1 + act = bf16-only output through LDS, 4 + act = fp32 output (+ residual, + bf16 shadow) through LDS.
// TR = the weight-gradient form dW = X^T dY: A16 is X (K, M) and B16 is dY (K, N), both with the contraction index as the SLOW
// dimension.  Items are then 64 k rows x 128 B (A_i: 64 m; B_j a | b: the 32 n of two waves side by side), copied as they lie in
// memory, and the k-contiguous MFMA fragments come out of the transposing LDS read ds_read_b64_tr_b16 (two per fragment), exactly
// as in gemm_bf16_tr_kernel -- same products, same order, identical bits.  Everything else (stream, ring, phases) is shared.
// 16-byte slot swizzle of a k-major image row.  A transposing read moves 32 lanes x 8 B per LDS pass: four k rows (128 B apart) x
// two 32-byte column runs.  Rows r and r + 2 would land on the same banks (LDS is 256 B wide: row parity is the only row bit in
// the bank index), so rows 2, 3 (mod 4) swap the two 64-byte halves of their 128 bytes -- slot ^= 4.  (First version: slot ^= 2,
// which only permuted the runs inside a half: PMC showed half of the LDS cycles of this form as bank conflicts.)
__device__ __forceinline__ int sw_swzk(int row) { return ((row >> 1) & 1) << 2; }

template <int OFF>
__device__ __forceinline__ u32x2 sw_read_tr(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

template <bool TRACE, bool PRIO = true, int EK = 0, bool TR = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_sw_kernel(GemmSWArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sw_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = wc: the wave's 64-column group
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware tile order.  Blocks go to the 8 XCDs round-robin in dispatch order (x fastest, then z); the remap runs over the whole
    // (batch, tile) space so that every XCD owns ONE contiguous run of it whatever the tile count modulo 8 is, and inside a batch the
    // SHORTER grid dimension is the fastest index: a run then covers an r x c patch of tiles that share r + c operand panels in
    // that XCD's L2 instead of 1 + r c (forward shapes: N is the short side; weight gradients 768 x 3072: M).
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
    int tm = m_fast ? bid % g.tiles_m : bid / g.tiles_n, tn = m_fast ? bid / g.tiles_m : bid % g.tiles_n;
    if (!TR && g.gm > 0) grouped_tile(bid, g.tiles_m, g.tiles_n, g.gm, tm, tn);      // wide outputs: gm x (64 / gm) patches in flight (common.h)
    const int m0 = tm * SW_BM, n0 = tn * SW_BN;
    const int zk = TR ? (z < g.kextra ? z : g.kextra) : 0;                  // K tiles the earlier (longer) slabs pushed this one back by
    const int nk = g.K / SW_BK + ((TR && z < g.kextra) ? 1 : 0);
    const int kt_rag = (TR && g.krag > 0 && z == (int)gridDim.z - 1) ? nk - 1 : -1;      // the K tile whose tail rows do not exist (block-uniform)
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
#define SW_TRC() do { if (TRACE && trc && tid == 0 && trc_n < 31) trc[trc_n++] = clock64(); } while (0)
#else
#define SW_TRC() do { } while (0)
#endif

    // ---- LDS-DMA sources.  An item is 64 image rows = 8 pieces of 1 KiB, two per wave: piece pc = image rows 8 pc .. 8 pc + 7,
    // the lane at physical slot (lane & 7) of image row r fetches logical slot (lane & 7) ^ swz(r).  Per-lane byte offsets from a
    // scalar base: A per half (rows clamped to the matrix, so a ragged last row tile re-reads row M - 1), B once (whole column tiles:
    // the two halves and two column groups of B are scalar offsets of the base).
    uint32_t offA[2][2], offB[2];
    uint32_t offAr[2] = {0u, 0u}, offBr[2] = {0u, 0u};      // transposed form, ragged last K tile: rows clamped as GemmSWArgs::krag says
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + (lane >> 3);                       // image row 0 .. 63
        if constexpr (TR) {
            // image row = k; the 128 bytes of a row are 64 m of A (one run) or 32 n of each of two waves of B (two runs 64 columns apart)
            const uint32_t ls = (uint32_t)((lane & 7) ^ sw_swzk(r));          // logical 16-byte slot
            offA[0][i] = offA[1][i] = 2u * ((uint32_t)((int64_t)r * g.lda) + ls * 8u);
            offB[i] = 2u * ((uint32_t)((int64_t)r * g.ldb16) + (ls >> 2) * 64u + (ls & 3u) * 8u);
            const int ra = r < g.krag ? r : g.krag - 1, rb = r < g.krag ? r : g.krag;      // (the LDS position stays row r's: only the source moves)
            offAr[i] = 2u * ((uint32_t)((int64_t)ra * g.lda) + ls * 8u);
            offBr[i] = 2u * ((uint32_t)((int64_t)rb * g.ldb16) + (ls >> 2) * 64u + (ls & 3u) * 8u);
        } else {
            const uint32_t sl = (uint32_t)(((lane & 7) ^ sw_swz(r)) << 3);    // logical slot, in elements
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int ar = h * 64 + r;
                ar = m0 + ar < g.M ? ar : g.M - 1 - m0;
                offA[h][i] = 2u * ((uint32_t)((int64_t)ar * g.lda) + sl);
            }
            const int bc = (r >> 5) * 64 + (r & 31);                          // column of B_0a's image row r (B_1: + 32, half b: + 128)
            offB[i] = 2u * ((uint32_t)((int64_t)bc * g.ldb16) + sl);
        }
    }
    auto uniform_ptr = [](const uint16_t* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const unsigned char*>(((uint64_t)hi << 32) | lo);
    };
    const unsigned char* const baseA = uniform_ptr(TR ? g.A16 + (int64_t)z * g.strideA + (int64_t)zk * SW_BK * g.lda + m0
                                                      : g.A16 + (int64_t)z * g.strideA + (int64_t)m0 * g.lda);
    const unsigned char* const baseB = uniform_ptr(TR ? g.B16 + (int64_t)z * g.strideB + (int64_t)zk * SW_BK * g.ldb16 + n0 : g.B16 + (int64_t)n0 * g.ldb16);
    const int64_t bstep32 = TR ? 64 : 64 * g.ldb16, bstep128 = TR ? 256 : 256 * g.ldb16;      // bytes: 32 / 128 columns of B
    const int64_t kstepA = TR ? 2 * SW_BK * g.lda : 2 * SW_BK, kstepB = TR ? 2 * SW_BK * g.ldb16 : 2 * SW_BK;      // bytes per K tile
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)sw_smem;

    // piece I (0 | 1) of this wave's share of stream item J (= item number mod 6) of K tile `ktile`, into ring slot `slot`
    auto issue_piece = [&](auto Jc, auto Ic, int ktile, int slot) {
        constexpr int J = decltype(Jc)::value, I = decltype(Ic)::value;
        const unsigned dst = lds0 + (unsigned)slot * SW_ITEM + (unsigned)wave * 2048u + (unsigned)I * 1024u;
        if constexpr (J == 0 || J == 5) {
            const uint32_t off = (TR && ktile == kt_rag) ? offAr[I] : offA[J == 0 ? 0 : 1][I];
            sw_dma(dst, off, baseA + (int64_t)ktile * kstepA + ((TR && J == 5) ? 128 : 0));
        } else {
            constexpr int HALF = (J == 2 || J == 4) ? 1 : 0, JB = (J >= 3) ? 1 : 0;      // waves 2-3 | columns + 32
            const uint32_t off = (TR && ktile == kt_rag) ? offBr[I] : offB[I];
            sw_dma(dst, off, baseB + (int64_t)ktile * kstepB + HALF * bstep128 + JB * bstep32);
        }
    };
    auto issue = [&](auto Jc, int ktile, int slot) {
        issue_piece(Jc, IC<0>{}, ktile, slot);
        issue_piece(Jc, IC<1>{}, ktile, slot);
    };

    // ---- fragment reads.  Image row rho = 32 rb + li (A) or 32 (wave & 1) + li (B): swz(rho) = swz(li) for both, so one pair
    // (x, y) serves every read: address = slot base + x + ((32 ks) ^ y) [+ 4096 for the second 32 rows].
    const unsigned x0 = lds0 + (unsigned)li * 128u + 16u * (unsigned)(lh ^ (sw_swz(li) & 1)), y0 = (unsigned)(sw_swz(li) & 6) * 16u;
    unsigned xk[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xk[ks] = x0 + ((32u * ks) ^ y0);
    const unsigned bwave = (unsigned)(wave & 1) * 4096u;           // this wave's 32 rows inside its B item

    // transposed form: a lane supplies the address of 4 consecutive columns (8 B) of k row 8 lh + l16 / 4 (+ 4 for the second
    // read) and receives 4 k of column 16 grp + l16; slot swizzle = sw_swzk(row) = a per-lane constant (rows step by multiples of 4)
    const int l16 = lane & 15, grp = (lane >> 4) & 1;
    const unsigned xtr = lds0 + (unsigned)(8 * lh + (l16 >> 2)) * 128u + (unsigned)(((2 * grp + ((l16 >> 1) & 1)) ^ (4 * ((l16 >> 3) & 1))) << 4) +
                         8u * (unsigned)(l16 & 1);
    const unsigned xtr1 = xtr ^ 64u;      // the other 64-byte half of the row (A: the second 32 m; B: the odd wave's 32 n): XOR, because the swizzle swaps halves
    bf16x8 fa[2][2][4];      // [i: 64-row half][rb: 32-row block][ks]
    bf16x8 fb[2][4];         // [j: 32-column half][ks]
    u32x2 tal[2][2][4], tah[2][2][4], tbl[2][4], tbh[2][4];      // transposed form: the two 4-k halves of each fragment
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 zero = {};
        const u32x2 z2 = {0u, 0u};
        fa[0][0][ks] = fa[0][1][ks] = fa[1][0][ks] = fa[1][1][ks] = fb[0][ks] = fb[1][ks] = zero;
        tal[0][0][ks] = tal[0][1][ks] = tal[1][0][ks] = tal[1][1][ks] = tah[0][0][ks] = tah[0][1][ks] = tah[1][0][ks] = tah[1][1][ks] = z2;
        tbl[0][ks] = tbl[1][ks] = tbh[0][ks] = tbh[1][ks] = z2;
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ring bookkeeping (scalar): s0 = slot of item 6 kt (the K tile's A_0); slot of item 6 kt + j = (s0 + j) mod 10
    auto slot_of = [](int s0, int j) { const int s = s0 + j; return s >= SW_SLOTS ? s - SW_SLOTS : s; };
    auto rd_a = [&](auto Ic, auto KSc, int slot) {          // A_I, k step KS: both 32-row blocks
        constexpr int I = decltype(Ic)::value, KS = decltype(KSc)::value;
        if constexpr (TR) {
            const unsigned a = xtr + (unsigned)slot * SW_ITEM, a1 = xtr1 + (unsigned)slot * SW_ITEM;
            tal[I][0][KS] = sw_read_tr<KS * 2048>(a);
            tah[I][0][KS] = sw_read_tr<KS * 2048 + 512>(a);
            tal[I][1][KS] = sw_read_tr<KS * 2048>(a1);
            tah[I][1][KS] = sw_read_tr<KS * 2048 + 512>(a1);
        } else {
            const unsigned a = xk[KS] + (unsigned)slot * SW_ITEM;
            fa[I][0][KS] = sw_read<0>(a);
            fa[I][1][KS] = sw_read<4096>(a);
        }
    };
    auto rd_b = [&](auto Jc, auto KSc, int slot) {          // B_J, k step KS (slot = this wave's half a | b)
        constexpr int J = decltype(Jc)::value, KS = decltype(KSc)::value;
        if constexpr (TR) {
            const unsigned a = ((wave & 1) ? xtr1 : xtr) + (unsigned)slot * SW_ITEM;
            tbl[J][KS] = sw_read_tr<KS * 2048>(a);
            tbh[J][KS] = sw_read_tr<KS * 2048 + 512>(a);
        } else {
            fb[J][KS] = sw_read<0>(xk[KS] + (unsigned)slot * SW_ITEM + bwave);
        }
    };
    auto mm = [&](auto QIc, auto QJc, auto KSc) {
        constexpr int QI = decltype(QIc)::value, QJ = decltype(QJc)::value, KS = decltype(KSc)::value;
        if constexpr (TR) {
            union J8 { u32x2 h[2]; bf16x8 v; } a0, a1, b;
            a0.h[0] = tal[QI][0][KS]; a0.h[1] = tah[QI][0][KS];
            a1.h[0] = tal[QI][1][KS]; a1.h[1] = tah[QI][1][KS];
            b.h[0] = tbl[QJ][KS]; b.h[1] = tbh[QJ][KS];
            acc[QI * 2][QJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, b.v, acc[QI * 2][QJ], 0, 0, 0);
            acc[QI * 2 + 1][QJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b.v, acc[QI * 2 + 1][QJ], 0, 0, 0);
        } else {
            acc[QI * 2][QJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[QI][0][KS], fb[QJ][KS], acc[QI * 2][QJ], 0, 0, 0);
            acc[QI * 2 + 1][QJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[QI][1][KS], fb[QJ][KS], acc[QI * 2 + 1][QJ], 0, 0, 0);
        }
    };
#define SW_TIE24(P, X, Y)                                                                                                           \
    asm volatile(P : "+v"(X[0][0][0]), "+v"(X[0][0][1]), "+v"(X[0][0][2]), "+v"(X[0][0][3]), "+v"(X[0][1][0]), "+v"(X[0][1][1]),            \
                 "+v"(X[0][1][2]), "+v"(X[0][1][3]), "+v"(X[1][0][0]), "+v"(X[1][0][1]), "+v"(X[1][0][2]), "+v"(X[1][0][3]),                \
                 "+v"(X[1][1][0]), "+v"(X[1][1][1]), "+v"(X[1][1][2]), "+v"(X[1][1][3]), "+v"(Y[0][0]), "+v"(Y[0][1]),                      \
                 "+v"(Y[0][2]), "+v"(Y[0][3]), "+v"(Y[1][0]), "+v"(Y[1][1]), "+v"(Y[1][2]), "+v"(Y[1][3]) :: "memory")
    // wait for every fragment read in flight and pin the fragment registers behind the wait (the MFMAs cannot move above it)
#define SW_TIE_ALL()                                                                                                                \
    do {                                                                                                                            \
        if constexpr (TR) {                                                                                                         \
            SW_TIE24("s_waitcnt lgkmcnt(0)", tal, tbl);                                                                             \
            SW_TIE24("", tah, tbh);                                                                                                 \
        } else {                                                                                                                    \
            SW_TIE24("s_waitcnt lgkmcnt(0)", fa, fb);                                                                               \
        }                                                                                                                           \
    } while (0)

    // One phase of K tile kt (s0 = ring slot of its item 0).  Q: quadrant; VM: vmcnt before the barrier (-1: none); NI: items to
    // request (0 .. 2); READ: the items this phase reads exist.  What a phase requests is fixed by its position: the slots freed by
    // the reads of the phase before it, i.e. items 6 kt + 11, 12 | 13, 14 | 15 | 16 for Q = 0 | 1 | 2 | 3.
    auto phase = [&](auto Qc, auto VMc, auto NIc, auto READc, int s0, int kt) {
        constexpr int Q = decltype(Qc)::value, VM = decltype(VMc)::value, NI = decltype(NIc)::value;
        constexpr bool READ = decltype(READc)::value != 0;
        sw_wait_vm<VM>();
        SW_TIE_ALL();                                                 // reads issued during the previous phase (also: its slots are free)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // what this phase requests (two 1-KiB pieces per item and wave), threaded between the MFMA pairs below
        constexpr int J1 = Q == 0 ? 5 : Q == 1 ? 1 : Q == 2 ? 3 : 4, J2 = Q == 0 ? 0 : 2;      // items 6 kt + 11, 12 | 13, 14 | 15 | 16
        const int kt1 = Q == 0 ? kt + 1 : kt + 2, sl1 = slot_of(s0, Q == 0 ? 1 : Q == 1 ? 3 : Q == 2 ? 5 : 6), sl2 = slot_of(s0, Q == 0 ? 2 : 4);
        const int half = wave >> 1;                                   // this wave's B item of a pair: a (waves 0-1) | b (waves 2-3)
        // quadrant of this phase, and the registers its reads refill: Q0 (A_0, B_0) -> B_1 (items 3 | 4); Q1 (A_0, B_1) -> A_1 (item 5);
        // Q2 (A_1, B_1) -> A_0 of the next K tile (item 6); Q3 (A_1, B_0) -> B_0 of the next K tile (items 7 | 8), k step by k step
        // right behind the MFMAs that used the old values
        constexpr int QI = (Q == 0 || Q == 1) ? 0 : 1, QJ = (Q == 0 || Q == 3) ? 0 : 1;
        const int srd = slot_of(s0, Q == 0 ? 3 + half : Q == 1 ? 5 : Q == 2 ? 6 : 7 + half);
        mm(IC<QI>{}, IC<QJ>{}, IC<0>{});
        if constexpr (READ) {
            if constexpr (Q == 0) { rd_b(IC<1>{}, IC<0>{}, srd); rd_b(IC<1>{}, IC<1>{}, srd); rd_b(IC<1>{}, IC<2>{}, srd); rd_b(IC<1>{}, IC<3>{}, srd); }
            if constexpr (Q == 1) { rd_a(IC<1>{}, IC<0>{}, srd); rd_a(IC<1>{}, IC<1>{}, srd); rd_a(IC<1>{}, IC<2>{}, srd); rd_a(IC<1>{}, IC<3>{}, srd); }
            if constexpr (Q == 2) { rd_a(IC<0>{}, IC<0>{}, srd); rd_a(IC<0>{}, IC<1>{}, srd); rd_a(IC<0>{}, IC<2>{}, srd); rd_a(IC<0>{}, IC<3>{}, srd); }
            if constexpr (Q == 3) rd_b(IC<0>{}, IC<0>{}, srd);
        }
        if constexpr (NI >= 1) issue_piece(IC<J1>{}, IC<0>{}, kt1, sl1);
        __builtin_amdgcn_sched_barrier(0);
        mm(IC<QI>{}, IC<QJ>{}, IC<1>{});
        if constexpr (READ && Q == 3) rd_b(IC<0>{}, IC<1>{}, srd);
        if constexpr (NI >= 1) issue_piece(IC<J1>{}, IC<1>{}, kt1, sl1);
        __builtin_amdgcn_sched_barrier(0);
        mm(IC<QI>{}, IC<QJ>{}, IC<2>{});
        if constexpr (READ && Q == 3) rd_b(IC<0>{}, IC<2>{}, srd);
        if constexpr (NI >= 2) issue_piece(IC<J2>{}, IC<0>{}, kt + 2, sl2);
        __builtin_amdgcn_sched_barrier(0);
        mm(IC<QI>{}, IC<QJ>{}, IC<3>{});
        if constexpr (READ && Q == 3) rd_b(IC<0>{}, IC<3>{}, srd);
        if constexpr (NI >= 2) issue_piece(IC<J2>{}, IC<1>{}, kt + 2, sl2);
        __builtin_amdgcn_sched_barrier(0);
    };

    // (Round 6, measured dead end: starting the second block of every CU's first pair half a tile period late -- so that one block of a CU
    //  stores while the other has the matrix pipe, and the chip sees two half-size store bursts instead of one -- changed nothing:
    //  base fine-tune step 31.75 - 31.86 ms without, 31.54 - 32.10 ms with 25 / 50 / 75 % of a period; large 93.4 - 93.9 vs 93.5 - 93.9,
    //  profiles/r06_ab_gemm_stagger_*.txt.  The blocks do not stay in lock step long enough for it to matter.)
    // ---- prologue: ten items in flight; then the two read-only "phases" in front of K tile 0 (A_0, then B_0)
    issue(IC<0>{}, 0, 0); issue(IC<1>{}, 0, 1); issue(IC<2>{}, 0, 2); issue(IC<3>{}, 0, 3); issue(IC<4>{}, 0, 4); issue(IC<5>{}, 0, 5);
    issue(IC<0>{}, 1, 6); issue(IC<1>{}, 1, 7); issue(IC<2>{}, 1, 8); issue(IC<3>{}, 1, 9);
    sw_wait_vm<18>();                                                 // item 0 (nine younger items of two pieces each may be in flight)
    __builtin_amdgcn_s_barrier();
    rd_a(IC<0>{}, IC<0>{}, 0); rd_a(IC<0>{}, IC<1>{}, 0); rd_a(IC<0>{}, IC<2>{}, 0); rd_a(IC<0>{}, IC<3>{}, 0);
    sw_wait_vm<14>();                                                 // items 1, 2
    SW_TIE_ALL();
    __builtin_amdgcn_s_barrier();
    {
        const int sb = 1 + (wave >> 1);
        rd_b(IC<0>{}, IC<0>{}, sb); rd_b(IC<0>{}, IC<1>{}, sb); rd_b(IC<0>{}, IC<2>{}, sb); rd_b(IC<0>{}, IC<3>{}, sb);
    }
    issue(IC<4>{}, 1, 0);                                             // item 10 (slot 0: every wave's A_0 reads retired before the barrier)

    SW_TRC();
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);                // the K loop outranks the co-resident block's epilogue on this SIMD
    // ---- K loop.  Steady state: a phase requests as many items as the previous phase read (slots freed), and waits until the items
    // it reads itself have landed: 10 - reads(previous) - reads(this) younger items of two pieces each may stay in flight.
    int s0 = 0, kt = 0;
    for (; kt + 2 < nk; ++kt) {
        phase(IC<0>{}, IC<12>{}, IC<2>{}, IC<1>{}, s0, kt);
        phase(IC<1>{}, IC<14>{}, IC<2>{}, IC<1>{}, s0, kt);
        phase(IC<2>{}, IC<16>{}, IC<1>{}, IC<1>{}, s0, kt);
        phase(IC<3>{}, IC<14>{}, IC<1>{}, IC<1>{}, s0, kt);
        s0 = s0 + 6 >= SW_SLOTS ? s0 + 6 - SW_SLOTS : s0 + 6;
    }
    SW_TRC();
    // ---- the last two K tiles (items E - 12 .. E - 1): everything but the last item has been requested when they begin
    phase(IC<0>{}, IC<12>{}, IC<1>{}, IC<1>{}, s0, kt);                // requests item E - 1 (A_1 of the last K tile)
    phase(IC<1>{}, IC<12>{}, IC<0>{}, IC<1>{}, s0, kt);                // reads E - 7; E - 1 - (E - 7) = 6 younger
    phase(IC<2>{}, IC<10>{}, IC<0>{}, IC<1>{}, s0, kt);                // reads E - 6
    phase(IC<3>{}, IC<6>{}, IC<0>{}, IC<1>{}, s0, kt);                 // reads E - 5, E - 4
    s0 = s0 + 6 >= SW_SLOTS ? s0 + 6 - SW_SLOTS : s0 + 6;
    phase(IC<0>{}, IC<2>{}, IC<0>{}, IC<1>{}, s0, kt + 1);             // reads E - 3, E - 2
    phase(IC<1>{}, IC<0>{}, IC<0>{}, IC<1>{}, s0, kt + 1);             // reads E - 1
    phase(IC<2>{}, IC<-1>{}, IC<0>{}, IC<0>{}, s0, kt + 1);
    phase(IC<3>{}, IC<-1>{}, IC<0>{}, IC<0>{}, s0, kt + 1);
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();                                     // every wave is done with the ring: the LDS is free for the epilogue
    SW_TRC();
#ifdef W2V2_TUNING
    if (g.abl == 2) {
        if (acc[0][0][0] == 12345.678f && acc[3][1][5] == 1.0f) g.C16[0] = 1;
        return;
    }
#endif

    // ---- epilogue: bias -> act -> + residual -> fp32 store and / or bf16 shadow
    const int64_t tile_off = (int64_t)z * g.strideC + (int64_t)m0 * g.ldc + (n0 + wave * 64);
    const bool whole = g.M - m0 >= 128;                               // (block-uniform; ragged last row tiles take the register epilogue)
    const float* const bw = g.bias ? g.bias + (n0 + wave * 64) : nullptr;
    const unsigned wb = lds0 + (unsigned)wave * 16384u;
    if (!TR && EK >= 1 && EK <= 3 && whole) {
        sw_epilogue_bf16<EK - 1>(SW_NT(g), acc, g.C16 + tile_off, bw, (int)g.ldc, wb, lane);
    } else if (!TR && EK >= 4 && whole) {
        sw_epilogue_f32<EK - 4>(SW_NT(g), acc, g.C + tile_off, g.C16 ? g.C16 + tile_off : nullptr, g.residual ? g.residual + tile_off : nullptr, bw, (int)g.ldc, wb,
                                lane);
    } else {
        gemm_epilogue<4, 2, true>(acc, g.C ? g.C + tile_off : nullptr, g.C16 ? g.C16 + tile_off : nullptr, g.residual ? g.residual + tile_off : nullptr,
                                  bw, (int)g.ldc, g.M - m0, g.N - (n0 + wave * 64), g.act, li, lh);
    }
#ifdef W2V2_TUNING
    SW_TRC();
    if (TRACE && trc && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        trc[trc_n++] = clock64();
        trc[31] = (unsigned long long)trc_n;
    }
#endif
#undef SW_TRC
#undef SW_TIE_ALL
#undef SW_TIE24
}

template <bool TRACE, int EK>
int launch_sw(GemmSWArgs& g, dim3 grid, hipStream_t s) {
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_sw_kernel<TRACE, true, EK>), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS));
        attr_set = true;
    }
    W2V2_LAUNCH((gemm_bf16_sw_kernel<TRACE, true, EK>), grid, dim3(256), SW_LDS, s, g);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace

#ifdef W2V2_TUNING
extern unsigned long long* g_tune_trace;
#endif

// Shapes: both operands as aligned bf16 shadows, whole 256-column tiles (B's per-lane offsets are shared by its four items),
// K a multiple of 64 with at least 3 K tiles, any M.
bool gemm_bf16_sw_ok(int M, int N, int K, int64_t lda, int64_t ldb16, int64_t strideA) {
    return M >= 1 && N >= 256 && N % 256 == 0 && K % 64 == 0 && K >= 192 && lda % 8 == 0 && ldb16 % 8 == 0 && strideA % 8 == 0 &&
           128 * lda < (1 << 29) && 256 * ldb16 < (1 << 29);
}

// Weight-gradient form: C_z (M, N) = A16_z^T B16_z with A16 (K, M) rows lda apart and B16 (K, N) rows ldb apart, K rows per batch.
bool gemm_bf16_swtr_ok(int M, int N, int K, int64_t lda, int64_t ldb, int64_t strideA, int64_t strideB) {
    return M >= 128 && M % 128 == 0 && N >= 256 && N % 256 == 0 && K % 64 == 0 && K >= 192 && lda % 8 == 0 && ldb % 8 == 0 && strideA % 8 == 0 &&
           strideB % 8 == 0 && 64 * lda < (1 << 29) && 64 * ldb < (1 << 29);
}

int launch_gemm_bf16_swtr(const uint16_t* A16, int64_t lda, int64_t strideA, const uint16_t* B16, int64_t ldb, int64_t strideB, float* C,
                          int64_t ldc, int64_t strideC, int M, int N, int K, int nbatch, hipStream_t s, int kextra, int krag) {
    W2V2_REQUIRE(A16 && B16 && C && gemm_bf16_swtr_ok(M, N, K, lda, ldb, strideA, strideB), "gemm_bf16_swtr: unsupported operands");
    W2V2_REQUIRE(kextra >= 0 && kextra < nbatch && (kextra == 0 || (strideA == (int64_t)K * lda && strideB == (int64_t)K * ldb)),
                 "gemm_bf16_swtr: uneven slabs need back-to-back slabs (strides = K rows) and kextra < batch");
    GemmSWArgs g;
    g.A16 = A16; g.B16 = B16; g.C = C; g.C16 = nullptr; g.bias = nullptr; g.residual = nullptr;
    g.lda = lda; g.ldb16 = ldb; g.ldc = ldc; g.strideA = strideA; g.strideC = strideC; g.strideB = strideB;
    W2V2_REQUIRE(krag >= 0 && krag < 64 && (krag == 0 || nbatch == 1 || (strideA == (int64_t)K * lda && strideB == (int64_t)K * ldb)),
                 "gemm_bf16_swtr: a ragged last K tile needs back-to-back slabs and 0 < rows < 64");
    g.M = M; g.N = N; g.K = K; g.act = 0; g.kextra = kextra; g.krag = krag;
    g.tiles_m = M / SW_BM;
    g.tiles_n = N / SW_BN;
#ifdef W2V2_TUNING
    g.trace = nullptr; g.abl = 0; g.nt = 0;
#endif
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_sw_kernel<false, true, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS));
        attr_set = true;
    }
    W2V2_LAUNCH((gemm_bf16_sw_kernel<false, true, 0, true>), dim3(g.tiles_m * g.tiles_n, 1, nbatch), dim3(256), SW_LDS, s, g);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_gemm_bf16_sw(const uint16_t* A16, int64_t lda, int64_t strideA, const uint16_t* B16, int64_t ldb16, float* C, uint16_t* C16,
                        int64_t ldc, int64_t strideC, const float* bias, const float* residual, int M, int N, int K, int nbatch, int act,
                        hipStream_t s) {
    W2V2_REQUIRE(A16 && B16 && (C || C16) && gemm_bf16_sw_ok(M, N, K, lda, ldb16, strideA), "gemm_bf16_sw: unsupported operands");
    GemmSWArgs g;
    g.A16 = A16; g.B16 = B16; g.C = C; g.C16 = C16; g.bias = bias; g.residual = residual;
    g.lda = lda; g.ldb16 = ldb16; g.ldc = ldc; g.strideA = strideA; g.strideC = strideC; g.strideB = 0;
    g.M = M; g.N = N; g.K = K; g.act = act;
    g.tiles_m = (M + SW_BM - 1) / SW_BM;
    g.tiles_n = N / SW_BN;
    g.gm = nbatch == 1 ? tile_group_rows(g.tiles_m, g.tiles_n, (int64_t)SW_BM * K * 2, 64) : 0;      // (2 blocks x 32 CUs in flight per XCD)
    // which epilogue: bf16-only and fp32 outputs go through LDS when the strides allow 16-byte row pieces
    // (they issue 16-byte stores to C / C16 and 16-byte residual loads at z * strideC + row * ldc + 4 | 8 j: the bases and the batch
    //  stride must keep that alignment too, else the register epilogue -- ek 0, element-wise accesses -- takes the tile)
    auto al = [](const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    const bool lds16 = C16 && !C && !residual && ldc % 8 == 0 && strideC % 8 == 0 && al(C16, 16);
    const bool lds32 = C && ldc % 4 == 0 && strideC % 4 == 0 && al(C, 16) && (!residual || al(residual, 16)) && (!C16 || al(C16, 8));
    const int ek = lds16 ? 1 + act : lds32 ? 4 + act : 0;
    dim3 grid(g.tiles_m * g.tiles_n, 1, nbatch);
#ifdef W2V2_TUNING
    g.trace = g_tune_trace;
    g.abl = tune_int("W2V2_PP_ABL", 0);
    g.nt = tune_int("W2V2_SW_NT", SW_NT_DEFAULT);
    if (g.trace) {                                               // traced instances of the epilogues the model's large shapes use
        if (tune_int("W2V2_TRACE_EPI", 1) == 0) return launch_sw<true, 0>(g, grid, s);
        switch (ek) {
            case 1: return launch_sw<true, 1>(g, grid, s);
            case 2: return launch_sw<true, 2>(g, grid, s);
            case 4: return launch_sw<true, 4>(g, grid, s);
            default: return launch_sw<true, 0>(g, grid, s);
        }
    }
#endif
    switch (ek) {
        case 1: return launch_sw<false, 1>(g, grid, s);
        case 2: return launch_sw<false, 2>(g, grid, s);
        case 3: return launch_sw<false, 3>(g, grid, s);
        case 4: return launch_sw<false, 4>(g, grid, s);
        case 5: return launch_sw<false, 5>(g, grid, s);
        case 6: return launch_sw<false, 6>(g, grid, s);
        default: return launch_sw<false, 0>(g, grid, s);
    }
}

}  // namespace w2v2
