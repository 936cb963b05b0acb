// Fused self-attention on the bf16 matrix pipe (precision mode W2V2_PRECISION_BF16, head size 64).
//
// Same algorithm and the same "transposed tile" layout trick as attention.hip -- S^T = K Q^T so that a lane
// owns one query column, online softmax in fp32 registers, O^T = V^T P^T taking P straight from the S^T
// accumulator registers -- with v_mfma_f32_32x32x16_bf16 doing the contractions.  q, k, v (and dO in the
// backward) come as bf16 SHADOWS written by the producing GEMM's epilogue (nearest-even roundings of the fp32
// values); products are exact, accumulation, max / exp / sum and the outputs stay fp32.
//
// Data movement (round 2; measured: the register-staged fp32 version spent 35 % of its time on staging):
//   * every streamed 64-row tile ([key][d] of K and V; [query][d] of Q and dO in the dK/dV kernel) goes from HBM / L2
//     to LDS by LDS-DMA (global_load_lds, 16 bytes per lane, no registers, no VALU), ONE row-major image per
//     tensor, double buffered, one barrier per tile;
//   * contractions over d read it with ds_read_b128 (8 consecutive d of a row = the MFMA's k run);
//   * contractions over the streamed rows (O^T += V^T P^T, dQ^T += K^T dS^T, dV^T += dO^T P, dK^T += Q^T dS) read the
//     SAME image with ds_read_b64_tr_b16: within a 16-lane group lane l supplies 8 bytes (4 d) of row l / 4 and
//     receives rows 0..3 of column l, which are exactly the 4-key runs {0..3} + 4 lh (+ 8) that the accumulator
//     registers 8h .. 8h+7 of the S^T tile hold -- so P / dS go from the accumulators to the B operand by pairwise
//     packing and no transposed copy of any tile exists;
//   * one 16-byte-slot XOR swizzle, applied on the DMA's SOURCE address (the DMA writes LDS lane-linearly), makes
//     both read patterns bank-conflict free:  slot ^= 4 ((row >> 1) & 1) + ((row >> 2) & 3).
// The scale d^-0.5 = 2^-3 is folded into the exponent (scores stay unscaled in the accumulators): scaling by a power
// of two commutes with every rounding involved, so the results are those of pre-scaled q.
// Blocks are mapped so that the query blocks of one (sample, head) run on the same XCD and share K / V in its L2.
#include <type_traits>

#include "common.h"
#include "train.h"

namespace w2v2 {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using v4s = __attribute__((ext_vector_type(4))) short;

constexpr int DH = 64;    // head size (768 / 12 = 1024 / 16 = 64 for every published checkpoint)
constexpr int KT = 64;    // streamed rows per tile
constexpr int NW = 4;     // waves per block, 32 owned rows (queries; keys in the dK/dV kernel) each
constexpr int ROWB = 128; // bytes per LDS row: 64 bf16
constexpr int IMG = KT * ROWB;
constexpr float LOG2E = 1.44269504088896340736f;
constexpr float SCALE = 0.125f;                 // DH^-0.5
constexpr float C2 = SCALE * LOG2E;             // exponent factor on UNSCALED scores
constexpr float MASK_BIAS = -10000.0f / SCALE;  // (1 - mask) * -10000 (encoder.py:256-257) in unscaled-score units

struct Attn16Args {
    const uint16_t* qkv16;      // (B, T, 3H) bf16: q | k | v
    const int32_t* frame_len;   // (B) or null
    float* ctx;                 // (B, T, H)
    uint16_t* ctx16;            // optional bf16 shadow of ctx (the out-projection GEMM's A operand)
    int B, T, H, heads;
    int nqb, nwork;             // row blocks per (sample, head); blocks in the grid
};

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// 16-byte-slot swizzle of a row-major 128-byte-row image (see the header)
__device__ __forceinline__ int swz(int row) { return 4 * ((row >> 1) & 1) + ((row >> 2) & 3); }

// Keep-bit words of the attention-probability dropout (AttnTrain::keep_bits): word [bh][key tile][lh][query], query padded
// to whole 128-row blocks (Tq = 128 nqb), key tiles padded to Tq / 64.  A lane of the forward holds 32 scores of its query per key
// tile, accumulator register r of sub-tile kt = key  64 tile + 32 kt + (r & 3) + 8 (r >> 2) + 4 lh; registers (2j, 2j + 1) are the
// even / odd element of ONE hash word and one packed bf16 pair.  Pair p = 8 kt + j owns bit p (even element, register 2j) and bit
// 16 + p (odd element, register 2j + 1): the packed keep mask of the pair (0xFFFF per kept half, train.h) ANDed with
// (1 << p) | (1 << (16 + p)) is the pair's contribution, and  (word << (15 - p))  puts both decisions on the sign bits of the two
// 16-bit halves again (keep_pair_mask).
__device__ __forceinline__ constexpr int keep_bit(int kt, int r) { return (r & 1) * 16 + 8 * kt + (r >> 1); }
__device__ __forceinline__ uint32_t keep_pair_mask(uint32_t bits, int pair) {      // packed 0xFFFF / 0 masks of pair `pair`
    w2v2_short2 d = __builtin_bit_cast(w2v2_short2, bits << (15 - pair));
    d = d >> 15;
    return __builtin_bit_cast(uint32_t, d);
}
__device__ __forceinline__ float keep_f32(float v, uint32_t bits, int bit) {         // v where `bit` of bits is set, else +0
    // v_bfe_i32 (a 1-bit signed field = all ones / zero) + v_and: written with the builtin because the generic shift form is folded
    // back into test + compare + select (three instructions) by the optimiser
    return __uint_as_float(__float_as_uint(v) & (uint32_t)__builtin_amdgcn_sbfe((int)bits, (unsigned)bit, 1u));
}
__device__ __forceinline__ int64_t keep_word(int bh, int tile, int lh, int nqb) {
    const int Tq = nqb * NW * 32;
    return (((int64_t)bh * (Tq / KT) + tile) * 2 + lh) * Tq;
}

// blockIdx -> work item such that one XCD (blockIdx % 8) walks consecutive items: the row blocks of one (sample, head)
__device__ __forceinline__ int xcd_work(int bid, int nwork) {
    const int q = nwork >> 3, r = nwork & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- LDS-DMA of one 64-row x 64-d bf16 tile: 8 pieces of 1 KiB (8 rows), 2 per wave ----
// lane -> row 8 p + lane / 8, physical slot lane % 8, which holds logical slot (lane % 8) ^ swz(row).
// Rows beyond `rows` are clamped (their results are masked or never stored).
struct TileDma {
    int r0, r1;       // this lane's two rows inside a tile
    int c0, c1;       // element offset of its 16-byte chunk in those rows
    __device__ __forceinline__ void init(int wave, int lane) {
        r0 = 16 * wave + (lane >> 3);
        r1 = r0 + 8;
        c0 = 8 * ((lane & 7) ^ swz(r0));
        c1 = 8 * ((lane & 7) ^ swz(r1));
    }
    // src: first row of the tensor for this (sample, head); ld in elements; img: LDS image base (wave-uniform)
    __device__ __forceinline__ void issue(const uint16_t* src, int ld, int row0, int rows, unsigned char* img, int wave) const {
        const int a0 = min(row0 + r0, rows - 1) * ld + c0, a1 = min(row0 + r1, rows - 1) * ld + c1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + a0),
                                         (__attribute__((address_space(3))) void*)(img + (2 * wave) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + a1),
                                         (__attribute__((address_space(3))) void*)(img + (2 * wave + 1) * 1024), 16, 0, 0);
    }
};

// ---- fragment reads ----
// A fragment of a contraction over d: 8 consecutive d of `row` (= sub * 32 + li), k-step st, lane half lh
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* img, int row, int xr, int st, int lh) {
    return as_bf16x8(*reinterpret_cast<const u32x4*>(img + row * ROWB + (((2 * st + lh) ^ xr) << 4)));
}
// A fragments of a contraction over the streamed rows: output row d = 32 dt + li, k = the rows that accumulator
// registers 8h .. 8h+7 of sub-tile t hold: 32 t + 16 h + 4 lh + {0..3} and + 8 + {0..3}.  Per-lane byte offsets
// (relative to row 32 t + 16 h) are precomputed once: lo/hi for dt = 0; dt = 1 flips slot bit 2 (^ 64 bytes).
struct TrOff {
    int lo, hi;
    __device__ __forceinline__ void init(int lane) {
        const int l16 = lane & 15, grp = (lane >> 4) & 1, lh = lane >> 5, q = l16 >> 2;
        const int slot = 2 * grp + ((l16 >> 1) & 1);
        lo = (4 * lh + q) * ROWB + ((slot ^ (4 * (q >> 1) + lh)) << 4) + 8 * (l16 & 1);
        hi = (4 * lh + q + 8) * ROWB + ((slot ^ (4 * (q >> 1) + lh + 2)) << 4) + 8 * (l16 & 1);
    }
};
__device__ __forceinline__ bf16x8 frag_cols(const unsigned char* img, const TrOff& o, int t, int h, int dt) {
    const unsigned char* p = img + (32 * t + 16 * h) * ROWB;
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + (o.lo ^ (dt << 6))));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + (o.hi ^ (dt << 6))));
    union { v4s h2[2]; bf16x8 v; } u;
    u.h2[0] = lo;
    u.h2[1] = hi;
    return u.v;
}
__device__ __forceinline__ bf16x8 pack_acc(const f32x16& v, int h) {
    u32x4 pb;
#pragma unroll
    for (int j = 0; j < 4; ++j) pb[j] = pack_bf16(v[8 * h + 2 * j], v[8 * h + 2 * j + 1]);
    return as_bf16x8(pb);
}
// [col][d] register fragments (B operand of the contractions over d): d = 16 st + 8 lh .. + 7 of one bf16 row
__device__ __forceinline__ void load_col_frags(u32x4 (&f)[4], const uint16_t* p) {
#pragma unroll
    for (int st = 0; st < 4; ++st) f[st] = *reinterpret_cast<const u32x4*>(p + 16 * st);
}

// ---- The forward, software-pipelined inside a wave (round 5; the plain two-stage kernel it replaced -- one barrier per 64-key tile, S
// MFMAs, softmax, PV MFMAs in sequence -- was removed in round 6, A/B in profiles/r05_ab_attention_pipe.txt).  Why: PMC of the plain kernel
// (profiles/r04_attention_bf16_pmc.md) showed 456 VALU instructions per 64-key tile against 16 MFMAs -- VALU-bound 3.5 : 1 -- yet the VALU
// pipe only 41 % busy: within a wave the S MFMAs wait for the tile barrier and their LDS reads, the softmax waits for the S MFMAs, every
// PV MFMA for its own transposing reads, and the four waves of a block walk those phases in lock step.  Here a wave works on 32-key
// SUB-tiles and always has the NEXT sub-tile's four S MFMAs (and their fragment reads) in flight under the softmax arithmetic of the
// current one:
//     even step:   [ S(tile j, keys 32..63) -> sb   ||  softmax(sa) ]   rescale O   [ PV(sa) ]
//     odd step:    barrier, DMA of tile j + 2       [ S(tile j + 1, keys 0..31) -> sa  ||  softmax(sb) ]   rescale O   [ PV(sb) ]
// The online softmax runs per sub-tile (running max / sum updated every 32 keys); O is rescaled only when some lane's maximum moved
// (a wave-uniform test: after the first tiles it almost never does).  K / V images sit in a three-slot ring: the odd step reads the
// next tile while the current tile's V is still needed, and the one barrier per tile both publishes tile j + 1 and frees tile j - 1's
// slot for the DMA of tile j + 2.  Same bf16 operands, fp32 scores / exponentials / sums as above; the probabilities are rounded to bf16
// relative to a running maximum that is updated every 32 keys.
// exp as one FMA-class op + v_exp_f32: exponent = s C2 - m C2 as ONE fused multiply-add (round 4; before: (s - m) * C2).  The product s C2 is
// exact inside the FMA; what is rounded is m C2, once per row: a relative error of 2^-24 |m C2| in every probability of the row -- 1e-6 at
// |m| = 100, against the 2^-9 the bf16 rounding of P commits next -- and the SAME factor in the row sum, so it cancels in the normalised
// output.  (A fully masked row, |m C2| = 1.4e4, would see 1e-3: such a row has no valid key and its output is discarded by the caller's mask.)
// Attention-probability dropout (encoder.py:42-44) is applied to the PACKED pairs: registers (2j, 2j + 1) of a sub-tile are the two halves
// of one hash word = one packed bf16 pair, so a word's two decisions (train.h::dropout_keep_mask_pk: two packed 16-bit instructions) mask
// the pair with one AND and enter the keep word with one bit-field insert; registers 8h .. 8h + 7 are one OCT of the hash (train.h: one
// mixer round and two 64-bit products for eight decisions; round 5 paid a full two-multiply mixer per pair, 60 % of this VALU-bound loop).
// The row sum uses the un-dropped p; 1 / (1 - p) is applied once, with the final normalisation.  What the forward saves for the backward
// kernels below is  lse2 = -(m C2 + log2 l)  -- minus the log-sum-exp in the exponent's own units, so that P = exp2(S C2 + lse2) is one
// fused multiply-add and one v_exp_f32 there.
template <bool TRAIN, bool DROP>
__global__ __launch_bounds__(NW * 64, 3) void attention_bf16_pipe_kernel(Attn16Args a, AttnTrain tr) {
    constexpr int STAGE = 2 * IMG;           // K image, V image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_a16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int work = xcd_work(blockIdx.x, a.nwork);
    const int bh = work / a.nqb, qb = work - bh * a.nqb;
    const int b = bh / a.heads, head = bh - b * a.heads;
    const int q0 = (qb * NW + wave) * 32;
    const int ld = 3 * a.H;
    const uint16_t* __restrict__ base = a.qkv16 + (int64_t)b * a.T * ld + head * DH;
    const int flen = a.frame_len ? a.frame_len[b] : a.T;
    const int ntiles = (a.T + KT - 1) / KT;

    TrOff tro;
    tro.init(lane);
    // (TileDma's addressing with the lane constants rebuilt per tile: two registers live across the loop instead of four)
    const int dma_r0 = 16 * wave + (lane >> 3), dma_l7 = lane & 7;
    auto issue = [&](int tile, int slot) {          // (tile clamped: past the end the last tile is fetched again, never used)
        unsigned char* S = smem_a16 + slot * STAGE + (2 * wave) * 1024;
        const int row0 = min(tile, ntiles - 1) * KT;
        const int a0 = min(row0 + dma_r0, a.T - 1) * ld + 8 * (dma_l7 ^ swz(dma_r0));
        const int a1 = min(row0 + dma_r0 + 8, a.T - 1) * ld + 8 * (dma_l7 ^ swz(dma_r0 + 8));
#pragma unroll
        for (int kv = 0; kv < 2; ++kv) {
            const uint16_t* src = base + (1 + kv) * a.H;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + a0),
                                             (__attribute__((address_space(3))) void*)(S + kv * IMG), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + a1),
                                             (__attribute__((address_space(3))) void*)(S + kv * IMG + 1024), 16, 0, 0);
        }
    };
    issue(0, 0);
    issue(1, 1);

    u32x4 qf[4];
    load_col_frags(qf, base + min(q0 + li, a.T - 1) * ld + 8 * lh);

    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const uint32_t drop_key = DROP ? dropout_key(tr.seed, tr.stream) : 0u, drop_thr = DROP ? dropout_threshold(tr.p) : 0u;
    const uint32_t thr1s_pk = dropout_thr1s_pk(drop_thr);
    const uint32_t drop_row = (uint32_t)(((uint64_t)bh * a.T + (uint64_t)min(q0 + li, a.T - 1)) * attention_drop_stride(a.T));
    constexpr bool drop = DROP;            // (a template parameter: as a run-time branch inside the loop the two PV paths kept O in
                                           //  different registers and the loop paid 16 v_mov_b64 + an MFMA drain per sub-tile to merge them)
    const int xr = swz(li);
    const int kend = min(flen, a.T);
    const uint32_t keep_lane = (uint32_t)(lh * a.nqb * NW * 32 + q0 + li);

    // S^T of one 32-key sub-tile: four MFMAs over d
    auto scores = [&](f32x16& s, const unsigned char* Ks, int kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st)
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ks, kt * 32 + li, xr, st, lh), as_bf16x8(qf[st]), s, 0, 0, 0);
    };
    // key-padding mask / tile padding on a sub-tile that touches the valid-length or T boundary (wave-uniform test by the caller)
    auto mask = [&](f32x16& s, int k0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float v = s[r];
            v = key >= flen ? v + MASK_BIAS : v;
            s[r] = key >= a.T ? -INFINITY : v;
        }
    };
    // online softmax of one sub-tile: s -> un-normalised probabilities relative to the new running maximum; returns the factor the
    // accumulated O and sum carry from the old maximum (1 when it did not move)
    auto softmax = [&](f32x16& s) -> float {
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * C2);       // exp2(-inf) = 0 on the first sub-tile
        const float mc = -m_new * C2;
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(s[r], C2, mc));
            s[r] = p;
            rs += p;
        }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        return alpha;
    };
    auto rescale = [&](float alpha) {
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0ull) {      // (wave-uniform: some lane's running maximum moved)
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
    };
    // O^T += V^T P^T for one sub-tile (packed pairs, dropout on the packed pairs).  pm0 = the lane's first oct of the tile, pre-multiplied
    auto pv = [&](f32x16& s, const unsigned char* Vs, auto ktc, uint32_t pm0, uint32_t& bits) {
        constexpr int kt = decltype(ktc)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 pw;
            uint32_t w[4] = {0u, 0u, 0u, 0u};
            if (DROP) {       // registers 8 h .. 8 h + 7 = columns 16 h + 8 lh + 0 .. 7 of the sub-tile = oct 4 kt + 2 h (+ lh, in pm0)
                const uint32_t x = dropout_oct_x(drop_key, pm0 + (uint32_t)(4 * kt + 2 * h) * DROPOUT_FIB);
                dropout_oct_words<0>(x, w[0], w[1]);
                dropout_oct_words<1>(x, w[2], w[3]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 8 * h + 2 * j;
                pw[j] = pack_bf16(s[r], s[r + 1]);
                if (DROP) {
                    const uint32_t km = dropout_keep_mask_pk(w[j], thr1s_pk);
                    pw[j] &= km;
                    constexpr uint32_t one = 1u;
                    const uint32_t sel = (one << (8 * kt + (r >> 1))) | (one << (16 + 8 * kt + (r >> 1)));      // keep_bit(kt, r), keep_bit(kt, r + 1)
                    // bits = (km & sel) | (bits & ~sel) as ONE bit-field insert with the selector in a scalar register (the compiler's own
                    // choice: 16 v_and with literal selectors + 8 v_or3 at the end of the tile, with all 16 masks live until then)
                    asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(bits) : "s"(sel), "v"(km));
                }
            }
            const bf16x8 pb = as_bf16x8(pw);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(Vs, tro, kt, h, dt), pb, o[dt], 0, 0, 0);
        }
    };

    f32x16 sa, sb;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // tile 0 (tile 1's four pieces may still fly)
    __syncthreads();
    scores(sa, smem_a16, 0);
    int slot = 0;                                              // ring slot of tile j
    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * KT;
        const unsigned char* Ks = smem_a16 + slot * STAGE;
        const unsigned char* Vs = Ks + IMG;
        const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
        const uint32_t pm0 = (((drop_row + (uint32_t)k0) >> 3) + (uint32_t)lh) * DROPOUT_FIB;      // (drop_row: a multiple of 16; k0: of 64)
        uint32_t bits = 0;
        // ---- even step: keys k0 .. k0 + 31
        if (k0 + 32 > kend) mask(sa, k0);
        scores(sb, Ks, 1);
        const float al0 = softmax(sa);
        rescale(al0);
        __builtin_amdgcn_sched_barrier(0);
        pv(sa, Vs, std::integral_constant<int, 0>{}, pm0, bits);
        // ---- odd step: keys k0 + 32 .. k0 + 63.  Tile j + 1 has landed (this wave's pieces: vmcnt(0); everyone's: the barrier), and every
        // wave is done with tile j - 1, whose slot takes the DMA of tile j + 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        issue(tile + 2, slot2);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 64 > kend) mask(sb, k0 + 32);
        scores(sa, smem_a16 + slot1 * STAGE, 0);
        const float al1 = softmax(sb);
        rescale(al1);
        __builtin_amdgcn_sched_barrier(0);
        pv(sb, Vs, std::integral_constant<int, 1>{}, pm0, bits);
        if (drop && tr.keep_bits) (tr.keep_bits + keep_word(bh, tile, 0, a.nqb))[keep_lane] = bits;     // (uniform base + 32-bit lane offset)
        slot = slot1;
    }

    const int q = q0 + li;
    if (TRAIN && q < a.T && lh == 0) tr.lse[(int64_t)bh * a.T + q] = -fmaf(m_run, C2, __builtin_amdgcn_logf(l_run));      // lse2 (v_log_f32 = log2)
    if (q < a.T) {
        const float inv = (drop ? 1.0f / (1.0f - tr.p) : 1.0f) / l_run;
        const int64_t o0 = ((int64_t)b * a.T + q) * a.H + head * DH + 4 * lh;
        if (a.ctx) {
            float* op = a.ctx + o0;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(op + 32 * d + 8 * g) =
                        f32x4{o[d][4 * g] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv};
        }
        if (a.ctx16) {
            uint16_t* hp = a.ctx16 + o0;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<u32x2*>(hp + 32 * d + 8 * g) =
                        u32x2{pack_bf16(o[d][4 * g] * inv, o[d][4 * g + 1] * inv), pack_bf16(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv)};
        }
    }
}

// ======================================================================================
// Backward (see attention.hip for the math).  Same ownership as the fp32 kernels -- a wave owns 32 queries
// (dQ) or 32 keys (dK, dV) whose [col][d] fragments sit in registers as B operands, the other side streams
// through LDS by LDS-DMA, one row-major image per tensor (header).  dS and P(dropped) go from the accumulator
// registers to the B operand by pairwise packing, exactly as P does in the forward.
// Operands rounded to bf16: q, k, v, dO (shadows), dS, P keep/(1-p).  fp32: scores, exp, D, dS arithmetic, sums.
// ======================================================================================
struct Attn16BwdArgs {
    const uint16_t* qkv16;  // (B, T, 3H)
    const int32_t* frame_len;
    const uint16_t* do16;   // (B, T, H) bf16 shadow of dO
    float* dvec;            // (B, heads, T): D = rowsum(dO o O); written by the dQ kernel when o16 / o32 is given, else an input
    const uint16_t* o16;    // optional O (B, T, H) as bf16 (the forward's ctx shadow) or as fp32 (rounded to bf16 on the way in): the dQ
    const float* o32;       // kernel then computes D itself from bf16(dO) and bf16(O) -- the same bits from either form
    float* dqkv;            // (B, T, 3H); may be null when dqkv16 is given
    uint16_t* dqkv16;       // optional bf16 shadow of dqkv (the A operand of the q|k|v data-gradient GEMM)
    float* colpart;         // optional (B nqb, 3H): per-block column sums of dqkv over the block's valid rows (-> the q|k|v bias gradient)
    int B, T, H, heads;
    int nqb, nwork;
};

constexpr int COLSUM_LDS = (NW * 32 * 65 + NW * 64) * 4;       // block_colsum's LDS footprint (bytes)

// out[d] (d < 64) = scale * sum over the block's valid rows of a [d][row] accumulator pair (row = lane li of each wave), in a fixed
// order.  All threads of the block must call it; `lds` is free scratch (the tile stages, after the last barrier of the loop).
__device__ __forceinline__ void block_colsum(const f32x16 (&acc)[2], bool row_ok, float scale, float* lds, int tid, float* out) {
    const int lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    float* W = lds + wave * (32 * 65) + li * 65 + 4 * lh;       // [row][d], rows padded to 65 floats: conflict-free both ways
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) W[32 * dt + (r & 3) + 8 * (r >> 2)] = row_ok ? acc[dt][r] * scale : 0.f;
    __syncthreads();
    const float* Wr = lds + wave * (32 * 65) + lane;             // thread (wave, d = lane): the wave's 32 rows of column d
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) sum += Wr[q * 65];
    float* P2 = lds + NW * 32 * 65;
    P2[tid] = sum;
    __syncthreads();
    if (tid < 64) out[tid] = (P2[tid] + P2[64 + tid]) + (P2[128 + tid] + P2[192 + tid]);
    __syncthreads();
}

// ---- dQ: block = 4 waves x 32 queries; streams 64-key tiles of K and V ----
template <bool BITS>       // BITS: the forward left its keep decisions in tr.keep_bits (p > 0)
__global__ __launch_bounds__(256, 3) void attention_bf16_bwd_dq_kernel(Attn16BwdArgs a, AttnTrain tr) {
    // K image, V image [, one keep word per lane and wave: DMA'd with the tile, one tile ahead -- round 4; round 3 loaded the word from
    // HBM at the top of the tile that uses it, and the wait-count pass's vmcnt(0) in front of the transposing reads made every tile
    // wait for that load]
    constexpr int STAGE = 2 * IMG + (BITS ? NW * 64 * 4 : 0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_a16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int work = xcd_work(blockIdx.x, a.nwork);
    const int bh = work / a.nqb, qb = work - bh * a.nqb;
    const int b = bh / a.heads, head = bh - b * a.heads;
    const int q0 = (qb * NW + wave) * 32;
    const int ld = 3 * a.H;
    const uint16_t* __restrict__ base = a.qkv16 + (int64_t)b * a.T * ld + head * DH;
    const int flen = a.frame_len ? a.frame_len[b] : a.T;
    const int qr = min(q0 + li, a.T - 1);
    const bool qok = q0 + li < a.T;

    TileDma dma;
    dma.init(wave, lane);
    TrOff tro;
    tro.init(lane);
    auto issue = [&](int tile, int buf) {
        unsigned char* S = smem_a16 + buf * STAGE;
        dma.issue(base + a.H, ld, tile * KT, a.T, S, wave);
        dma.issue(base + 2 * a.H, ld, tile * KT, a.T, S + IMG, wave);
        if (BITS)           // this lane's keep word of the tile: 4 bytes per lane, lane-linear into the wave's 256-byte slot
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tr.keep_bits + keep_word(bh, tile, lh, a.nqb) + q0 + li),
                                             (__attribute__((address_space(3))) void*)(S + 2 * IMG + wave * 256), 4, 0, 0);
    };
    issue(0, 0);

    u32x4 qf[4], dof[4];
    load_col_frags(qf, base + qr * ld + 8 * lh);
    load_col_frags(dof, a.do16 + ((int64_t)b * a.T + qr) * a.H + head * DH + 8 * lh);
    const int64_t sidx = (int64_t)bh * a.T + qr;
    const float nlse = tr.lse[sidx];          // lse2 = -(m C2 + log2 l), the forward's form
    float dv;
    if (a.o16 || a.o32) {         // D = sum_d dO O of this query: each lane half takes its 32 d; bf16 values (dO: the fragments just loaded), fp32 sums
        const int64_t r0 = ((int64_t)b * a.T + qr) * a.H + head * DH + 8 * lh;
        float acc = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            u32x4 ow;
            if (a.o16) {
                ow = *reinterpret_cast<const u32x4*>(a.o16 + r0 + 16 * st);
            } else {
                const f32x4 o0 = *reinterpret_cast<const f32x4*>(a.o32 + r0 + 16 * st), o1 = *reinterpret_cast<const f32x4*>(a.o32 + r0 + 16 * st + 4);
                ow = u32x4{pack_bf16(o0[0], o0[1]), pack_bf16(o0[2], o0[3]), pack_bf16(o1[0], o1[1]), pack_bf16(o1[2], o1[3])};
            }
            const u32x4 gw = dof[st];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float oa = __uint_as_float(ow[2 * j] << 16), ob = __uint_as_float(ow[2 * j] & 0xFFFF0000u);
                const float oc = __uint_as_float(ow[2 * j + 1] << 16), od = __uint_as_float(ow[2 * j + 1] & 0xFFFF0000u);
                const float ga = __uint_as_float(gw[2 * j] << 16), gb = __uint_as_float(gw[2 * j] & 0xFFFF0000u);
                const float gc = __uint_as_float(gw[2 * j + 1] << 16), gd = __uint_as_float(gw[2 * j + 1] & 0xFFFF0000u);
                acc += (oa * ga + ob * gb) + (oc * gc + od * gd);
            }
        }
        dv = acc + __shfl_xor(acc, 32, 64);
        if (qok && lh == 0) a.dvec[sidx] = dv;      // (the dK/dV kernel, launched next, reads it)
    } else {
        dv = a.dvec[sidx];
    }
    const uint32_t drop_key = dropout_key(tr.seed, tr.stream), drop_thr = dropout_threshold(tr.p);
    const float inv = drop_thr != 0u ? 1.0f / (1.0f - tr.p) : 1.0f;
    const uint64_t rowbase = (uint64_t)sidx * attention_drop_stride(a.T);
    const int xr = swz(li);

    f32x16 dq[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;

    const int ntiles = (a.T + KT - 1) / KT;
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * KT, buf = tile & 1;
        issue(min(tile + 1, ntiles - 1), buf ^ 1);      // unconditional: see attention.hip (a branch here costs the DMA / compute overlap)
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* Ks = smem_a16 + buf * STAGE;
        const unsigned char* Vs = Ks + IMG;
        uint32_t bits = 0;
        if (BITS) bits = reinterpret_cast<const uint32_t*>(Ks + 2 * IMG)[wave * 64 + lane];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
            const int row = kt * 32 + li;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ks, row, xr, st, lh), as_bf16x8(qf[st]), s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Vs, row, xr, st, lh), as_bf16x8(dof[st]), dp, 0, 0, 0);
            }
            // dS^T = P^T * (dP^T * keep/(1-p) - D);  P = exp(scale S - lse) = exp2(S C2 + lse2)
            if (k0 + KT > min(flen, a.T)) {         // boundary tiles only (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float sv = s[r];
                    sv = key >= flen ? sv + MASK_BIAS : sv;
                    s[r] = key < a.T ? sv : -INFINITY;       // exp2(-inf) = 0: a padding key contributes nothing
                }
            }
            const uint32_t cbase = (uint32_t)rowbase + (uint32_t)(k0 + kt * 32 + 8 * lh);      // attention_drop_col of the lane's keys: 16 a + 8 lh + 4 b + c
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], C2, nlse));
                float g = dp[r];
                if (BITS) g = keep_f32(g, bits, keep_bit(kt, r));
                else if (drop_thr != 0u) g = dropout_keep32(drop_key, cbase + (uint32_t)((r & 3) + 4 * ((r >> 2) & 1) + 16 * (r >> 3)), drop_thr) ? g : 0.f;
                s[r] = pv * fmaf(g, inv, -dv);
            }
            // dQ^T[d][q] += sum_key K^T[d][key] dS^T[key][q]
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x8 pb = pack_acc(s, h);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(Ks, tro, kt, h, dt), pb, dq[dt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    if (qok) {
        const int64_t o0 = ((int64_t)b * a.T + q0 + li) * ld + head * DH + 4 * lh;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = f32x4{dq[d][4 * g] * SCALE, dq[d][4 * g + 1] * SCALE, dq[d][4 * g + 2] * SCALE, dq[d][4 * g + 3] * SCALE};
                if (a.dqkv) *reinterpret_cast<f32x4*>(a.dqkv + o0 + 32 * d + 8 * g) = v;
                if (a.dqkv16)
                    *reinterpret_cast<u32x2*>(a.dqkv16 + o0 + 32 * d + 8 * g) = u32x2{pack_bf16_rne(v[0], v[1]), pack_bf16_rne(v[2], v[3])};
            }
    }
    if (a.colpart)
        block_colsum(dq, qok, SCALE, reinterpret_cast<float*>(smem_a16), tid, a.colpart + ((int64_t)b * a.nqb + qb) * ld + head * DH);
}

// ---- dK, dV: block = 4 waves x 32 keys; streams 64-query tiles of Q and dO (+ lse, D as two 64-float rows) ----
// (two waves per SIMD: 228 VGPRs, no scratch.  Held to three -- 168 VGPRs, 37 of them spilled inside the tile loop -- the kernel is 1.6 x
//  slower: attention 5.32 -> 6.99 ms per step on base, 24.2 -> 32.6 on large-robust, profiles/r05_ab_attention_dkv_3waves.txt)
template <bool BITS>
__global__ __launch_bounds__(256, 2) void attention_bf16_bwd_dkv_kernel(Attn16BwdArgs a, AttnTrain tr) {
    constexpr int WROW = KT + 4;      // keep words per (wave, lh) row, padded: the two rows a lane group reads land on different banks
    constexpr int STAGE = 2 * IMG + 2 * KT * 4 + (BITS ? NW * 2 * WROW * 4 : 0);      // Q, dO images; lse, D; per wave: 2 rows of keep words
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_a16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int work = xcd_work(blockIdx.x, a.nwork);
    const int bh = work / a.nqb, kb = work - bh * a.nqb;
    const int b = bh / a.heads, head = bh - b * a.heads;
    const int c0 = (kb * NW + wave) * 32;          // this wave's 32 keys
    const int ld = 3 * a.H;
    const uint16_t* __restrict__ base = a.qkv16 + (int64_t)b * a.T * ld + head * DH;
    const uint16_t* __restrict__ dobase = a.do16 + (int64_t)b * a.T * a.H + head * DH;
    const float* __restrict__ lsebase = tr.lse + (int64_t)bh * a.T;
    const float* __restrict__ dvbase = a.dvec + (int64_t)bh * a.T;
    const int flen = a.frame_len ? a.frame_len[b] : a.T;
    const int key = c0 + li;
    const int kr = min(key, a.T - 1);
    const bool kok = key < a.T;
    const float kmask = key >= flen ? -10000.0f * LOG2E : 0.0f;      // the key's mask bias, in exponent units
    const uint32_t drop_key = dropout_key(tr.seed, tr.stream), drop_thr = dropout_threshold(tr.p);
    // keep words (BITS): this wave's 32 keys are sub-tile kt = (c0 / 32) & 1 of key tile c0 / 64; lane li's key is register
    // (li & 3) + 4 (li >> 3) of the forward lane half (li >> 2) & 1
    const int Tq = a.nqb * NW * 32;
    const uint32_t* __restrict__ kbits = BITS ? tr.keep_bits + keep_word(bh, c0 / KT, 0, a.nqb) : nullptr;
    const int kb_half = (li >> 2) & 1, kb_bit = keep_bit((c0 >> 5) & 1, (li & 3) + 4 * (li >> 3));

    TileDma dma;
    dma.init(wave, lane);
    TrOff tro;
    tro.init(lane);
    auto issue = [&](int tile, int buf) {
        unsigned char* S = smem_a16 + buf * STAGE;
        dma.issue(base, ld, tile * KT, a.T, S, wave);
        dma.issue(dobase, a.H, tile * KT, a.T, S + IMG, wave);
        if (wave < 2) {         // lse (wave 0) and D (wave 1) of the tile's 64 queries: one 4-byte DMA per lane
            const float* src = (wave == 0 ? lsebase : dvbase) + min(tile * KT + lane, a.T - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(S + 2 * IMG + wave * KT * 4), 4, 0, 0);
        }
        if (BITS) {             // the keep words of (these 64 queries) x (this wave's key tile), both halves lh
#pragma unroll
            for (int f = 0; f < 2; ++f)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbits + f * Tq + tile * KT + lane),
                                                 (__attribute__((address_space(3))) void*)(S + 2 * IMG + 2 * KT * 4 + (2 * wave + f) * WROW * 4), 4,
                                                 0, 0);
        }
    };
    issue(0, 0);

    u32x4 kf[4], vf[4];
    load_col_frags(kf, base + kr * ld + a.H + 8 * lh);
    load_col_frags(vf, base + kr * ld + 2 * a.H + 8 * lh);
    const float inv = drop_thr != 0u ? 1.0f / (1.0f - tr.p) : 1.0f;
    const uint32_t drop_col = (uint32_t)((uint64_t)bh * a.T * attention_drop_stride(a.T)) + attention_drop_col((uint32_t)kr);
    const int xr = swz(li);

    f32x16 dk[2], dvv[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[d][r] = dvv[d][r] = 0.f;

    const int ntiles = (a.T + KT - 1) / KT;
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int t0 = tile * KT, buf = tile & 1;
        issue(min(tile + 1, ntiles - 1), buf ^ 1);      // unconditional: see attention.hip (a branch here costs the DMA / compute overlap)
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* Qs = smem_a16 + buf * STAGE;
        const unsigned char* Os = Qs + IMG;
        const float* Ls = reinterpret_cast<const float*>(Qs + 2 * IMG);      // [0, KT): lse2, [KT, 2KT): D
        const uint32_t* Ws = reinterpret_cast<const uint32_t*>(Qs + 2 * IMG + 2 * KT * 4) + (2 * wave + kb_half) * WROW;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
            const int row = qt * 32 + li;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Qs, row, xr, st, lh), as_bf16x8(kf[st]), s, 0, 0, 0);     // S[q][key]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Os, row, xr, st, lh), as_bf16x8(vf[st]), dp, 0, 0, 0);   // dP[q][key]
            }
            // lane owns key column `key`; register r is query row qt*32 + (r&3) + 8 (r>>2) + 4 lh
            // (columns of keys >= T are clamped duplicates whose results are never stored, so only the QUERY bound
            // needs masking, and only on the last tile)
            uint32_t didx = drop_col + (uint32_t)(t0 + qt * 32 + 4 * lh) * attention_drop_stride(a.T);   // ((bh T + q) T + key) mod 2^32
            // P[q][key] = exp2(S C2 + lse2[q] (+ the key's mask bias)): one fused multiply-add per score where no key of the wave is masked
            if (c0 + 32 <= flen) {          // (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], C2, Ls[qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh]));
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], C2, kmask) + Ls[qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh]);
            }
            if (t0 + KT > a.T) {          // the last tile only (wave-uniform): query rows past T contribute nothing
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = t0 + qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh < a.T ? s[r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float pv = s[r];
                float g = dp[r], pd = pv;             // (Pd's factor 1 / (1 - p) multiplies dV once, at the end -- as the forward does with O)
                if (BITS) {             // one signed 1-bit field extract (v_bfe_i32) turns this lane's bit of the row's keep word into a mask for both values
                    const uint32_t km = (uint32_t)__builtin_amdgcn_sbfe((int)Ws[ql], (unsigned)kb_bit, 1u);
                    g = __uint_as_float(__float_as_uint(g) & km);
                    pd = __uint_as_float(__float_as_uint(pd) & km);
                } else if (drop_thr != 0u) {
                    const bool keep = dropout_keep32(drop_key, didx + (uint32_t)((r & 3) + 8 * (r >> 2)) * attention_drop_stride(a.T), drop_thr);
                    g = keep ? g : 0.f;
                    pd = keep ? pd : 0.f;
                }
                dp[r] = pd;                                    // Pd[q][key]
                s[r] = pv * fmaf(g, inv, -Ls[KT + ql]);        // dS[q][key]
            }
            // dV^T[d][key] += sum_q dO^T[d][q] Pd[q][key];   dK^T[d][key] += sum_q Q^T[d][q] dS[q][key]
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x8 pbp = pack_acc(dp, h), pbs = pack_acc(s, h);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dvv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(Os, tro, qt, h, dt), pbp, dvv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(Qs, tro, qt, h, dt), pbs, dk[dt], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    if (kok) {
        const int64_t k0 = ((int64_t)b * a.T + key) * ld + a.H + head * DH + 4 * lh;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // S was the UNSCALED score, so dS is the gradient of the scaled one: dK = scale * dS^T Q
                const f32x4 kv = f32x4{dk[d][4 * g] * SCALE, dk[d][4 * g + 1] * SCALE, dk[d][4 * g + 2] * SCALE, dk[d][4 * g + 3] * SCALE};
                const f32x4 vv = f32x4{dvv[d][4 * g] * inv, dvv[d][4 * g + 1] * inv, dvv[d][4 * g + 2] * inv, dvv[d][4 * g + 3] * inv};
                if (a.dqkv) {
                    *reinterpret_cast<f32x4*>(a.dqkv + k0 + 32 * d + 8 * g) = kv;
                    *reinterpret_cast<f32x4*>(a.dqkv + k0 + a.H + 32 * d + 8 * g) = vv;
                }
                if (a.dqkv16) {
                    *reinterpret_cast<u32x2*>(a.dqkv16 + k0 + 32 * d + 8 * g) = u32x2{pack_bf16_rne(kv[0], kv[1]), pack_bf16_rne(kv[2], kv[3])};
                    *reinterpret_cast<u32x2*>(a.dqkv16 + k0 + a.H + 32 * d + 8 * g) = u32x2{pack_bf16_rne(vv[0], vv[1]), pack_bf16_rne(vv[2], vv[3])};
                }
            }
    }
    if (a.colpart) {
        float* cp = a.colpart + ((int64_t)b * a.nqb + kb) * ld + head * DH;
        block_colsum(dk, kok, SCALE, reinterpret_cast<float*>(smem_a16), tid, cp + a.H);
        block_colsum(dvv, kok, inv, reinterpret_cast<float*>(smem_a16), tid, cp + 2 * a.H);
    }
}

// a bf16 shadow the caller did not bring: round the fp32 tensor into per-stream scratch (the same roundings the producing
// GEMM's epilogue would have written, so results do not depend on who made the shadow)
int shadow_or_scratch(const float* x, const uint16_t* x16, int64_t n, int slot, hipStream_t s, const uint16_t** out) {
    if (x16) {
        *out = x16;
        return W2V2_OK;
    }
    W2V2_REQUIRE(x, "attention_bf16: neither the fp32 tensor nor its bf16 shadow was given");
    void* p = nullptr;
    if (int e = stream_scratch(slot, s, (size_t)n * sizeof(uint16_t), &p)) return e;
    if (int e = launch_to_bf16(x, reinterpret_cast<uint16_t*>(p), n, s)) return e;
    *out = reinterpret_cast<const uint16_t*>(p);
    return W2V2_OK;
}

}  // namespace

bool attention_bf16_supported(int head_size) { return head_size == DH; }

int attention_colpart_rows(int B, int T) { return B * ((T + NW * 32 - 1) / (NW * 32)); }

int64_t attention_keep_bits_words(int B, int T, int heads) {
    const int64_t Tq = (int64_t)((T + NW * 32 - 1) / (NW * 32)) * NW * 32;
    return (int64_t)B * heads * (Tq / KT) * 2 * Tq;
}

// tr == nullptr: inference.  Otherwise the training forward (dropout on P, lse saved).
// qkv16: the bf16 shadow of qkv (null: made here from the fp32 tensor).  ctx or ctx16 may be null (not both).
int launch_attention_fwd_bf16(const float* qkv, const uint16_t* qkv16, const int32_t* frame_len, float* ctx, uint16_t* ctx16, int B,
                              int T, int H, int heads, const AttnTrain* tr, hipStream_t s) {
    W2V2_REQUIRE(H / heads == DH && H % heads == 0, "attention_bf16: head size %d unsupported (64)", H / heads);
    W2V2_REQUIRE(ctx || ctx16, "attention_bf16: no output");
    W2V2_REQUIRE((int64_t)T * 3 * H < (1ll << 31), "attention_bf16: T x 3H too large for 32-bit row offsets");
    // (the forward steps the hash's pair index inside a key tile by addition: equal to "(index mod 2^32) >> 1" while no index wraps)
    W2V2_REQUIRE(!tr || tr->p <= 0.f || (int64_t)B * heads * T * ((T + 15) & ~15) < (1ll << 32),
                 "attention_bf16: dropout over 2^32 or more attention probabilities is not supported");
    const uint16_t* q16 = nullptr;
    if (int e = shadow_or_scratch(qkv, qkv16, (int64_t)B * T * 3 * H, SCRATCH_QKV16, s, &q16)) return e;
    W2V2_REQUIRE(((reinterpret_cast<uintptr_t>(q16) | reinterpret_cast<uintptr_t>(ctx)) & 15) == 0 && (reinterpret_cast<uintptr_t>(ctx16) & 7) == 0,
                 "attention_bf16: unaligned operand");
    const int nqb = (T + NW * 32 - 1) / (NW * 32);
    Attn16Args a{q16, frame_len, ctx, ctx16, B, T, H, heads, nqb, nqb * heads * B};
    dim3 grid(a.nwork), block(NW * 64);
    const size_t lds = 3 * 2 * IMG;          // three-slot K / V ring: 48 KiB
    if (tr && (uint32_t)((double)tr->p * 65536.0) != 0u)          // (dropout_threshold(p) on the host)
        W2V2_LAUNCH((attention_bf16_pipe_kernel<true, true>), grid, block, lds, s, a, *tr);
    else if (tr)
        W2V2_LAUNCH((attention_bf16_pipe_kernel<true, false>), grid, block, lds, s, a, *tr);
    else
        W2V2_LAUNCH((attention_bf16_pipe_kernel<false, false>), grid, block, lds, s, a, AttnTrain{0.f, 0, 0, nullptr});
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

// dvec: (B, heads, T) scratch for D = rowsum(dO o O).  With `ctx16` (the forward's bf16 O) or `ctx` (fp32 O, rounded to bf16 on the
// way in) the dQ kernel computes it from the bf16 values of dO and O; otherwise it must already hold D.  qkv16 / dctx16: bf16 shadows
// (null: made here from the fp32 tensors).
int launch_attention_bwd_bf16(const float* qkv, const uint16_t* qkv16, const int32_t* frame_len, const float* dctx, const uint16_t* dctx16,
                              float* dvec, float* dqkv, uint16_t* dqkv16, int B, int T, int H, int heads, const AttnTrain& tr,
                              hipStream_t s, float* colpart, const float* ctx, const uint16_t* ctx16) {
    W2V2_REQUIRE(dqkv || dqkv16, "attention_bwd_bf16: no output");
    W2V2_REQUIRE(H / heads == DH && H % heads == 0, "attention_bwd_bf16: head size %d unsupported (64)", H / heads);
    W2V2_REQUIRE((int64_t)T * 3 * H < (1ll << 31), "attention_bwd_bf16: T x 3H too large for 32-bit row offsets");
    const uint16_t *q16 = nullptr, *do16 = nullptr;
    if (int e = shadow_or_scratch(qkv, qkv16, (int64_t)B * T * 3 * H, SCRATCH_QKV16, s, &q16)) return e;
    if (int e = shadow_or_scratch(dctx, dctx16, (int64_t)B * T * H, SCRATCH_DCTX16, s, &do16)) return e;
    W2V2_REQUIRE(((reinterpret_cast<uintptr_t>(q16) | reinterpret_cast<uintptr_t>(do16) | reinterpret_cast<uintptr_t>(dqkv)) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(dqkv16) & 7) == 0,
                 "attention_bwd_bf16: unaligned operand");
    const int nqb = (T + NW * 32 - 1) / (NW * 32);
    W2V2_REQUIRE(((reinterpret_cast<uintptr_t>(ctx) | reinterpret_cast<uintptr_t>(ctx16)) & 15) == 0, "attention_bwd_bf16: unaligned ctx");
    W2V2_REQUIRE(dvec, "attention_bwd_bf16: null dvec");
    Attn16BwdArgs a{q16, frame_len, do16, dvec, ctx16, ctx16 ? nullptr : ctx, dqkv, dqkv16, colpart, B, T, H, heads, nqb, nqb * heads * B};
    const bool bits = tr.keep_bits && (uint32_t)((double)tr.p * 65536.0) != 0u;      // = dropout_threshold(p) != 0: the forward's predicate for writing the words
    size_t lds_q = 2 * (2 * IMG + (bits ? NW * 64 * 4 : 0)), lds_kv = 2 * (2 * IMG + 2 * KT * 4 + (bits ? NW * 2 * (KT + 4) * 4 : 0));
    if (colpart) {
        if (lds_q < (size_t)COLSUM_LDS) lds_q = COLSUM_LDS;
        if (lds_kv < (size_t)COLSUM_LDS) lds_kv = COLSUM_LDS;
    }
    dim3 grid(a.nwork), block(256);
    if (bits) {
        W2V2_LAUNCH(attention_bf16_bwd_dq_kernel<true>, grid, block, lds_q, s, a, tr);
        W2V2_LAUNCH(attention_bf16_bwd_dkv_kernel<true>, grid, block, lds_kv, s, a, tr);
    } else {
        W2V2_LAUNCH(attention_bf16_bwd_dq_kernel<false>, grid, block, lds_q, s, a, tr);
        W2V2_LAUNCH(attention_bf16_bwd_dkv_kernel<false>, grid, block, lds_kv, s, a, tr);
    }
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace w2v2
