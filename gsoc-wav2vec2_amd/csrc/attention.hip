// Fused multi-head self-attention context in fp32 on the matrix cores.
//
// Reference: TransformerAttention.call / get_context (encoder.py:22-47):
//   q = (x Wq + bq) * d_h^-0.5;  scores = q k^T (+ additive mask);  p = softmax(scores, -1);
//   ctx = p v;  heads merged back to (B, T, H).
// The mask the encoder builds (encoder.py:256-263) is a key-padding mask:
// -10000 added to every score whose KEY frame is >= the row's valid frame count.
//
// TF materialises the (B, h, T, T) scores; here they never leave registers
// (online softmax over 64-key tiles).  Layout trick (wave64, v_mfma_f32_32x32x2_f32):
// compute the TRANSPOSED tile  S^T = K Q^T, so that in the MFMA C/D layout a lane
// owns ONE query column (lane & 31) and 16 of the 32 key rows; the softmax
// row-reduction is then 16 in-register values plus one exchange with lane ^ 32.
// The same registers are directly the B operand of  O^T = V^T P^T : the key index
// a lane holds in step r is exactly the key its half-wave must supply, because
// the k-pairing inside a 32x32x2 MFMA is free as long as A (V^T) uses the same
// key assignment.  No LDS round trip, no transposes, no cross-lane traffic
// beyond the one max/sum exchange.
//
// Block = NW waves x 32 queries of one (batch, head) (NW in {4, 6, 8, 12}, picked per launch to fill
// the 256 CUs in whole rounds); K and V tiles of 64 keys reach LDS by LDS-DMA, double-buffered, one
// barrier per tile, and are shared by the waves.
#include <limits.h>
#include <stdlib.h>

#include "common.h"
#include "train.h"

namespace w2v2 {

using f32x16 = __attribute__((ext_vector_type(16))) float;
// LDS fragment reads go through this NATIVE vector type, not HIP's float4 struct: after a struct-typed LDS load the compiler's
// wait-count pass assumes it may alias the LDS-DMA in flight and drains vmcnt(0) before the first read of every tile (measured:
// the fp32 attention forward waited for the next tile's K / V before touching the current one, 6.05 ms per forward).
using f32x4 = __attribute__((ext_vector_type(4))) float;

namespace {

constexpr int KT = 64;    // keys per tile

// exp(x) as v_exp_f32(x log2 e) with the rounding error of the product carried into a first-order correction:
// x log2e = y + e exactly (FMA residual + the low part of log2 e), exp(x) = 2^y (1 + e ln 2).  7 VALU ops against ~12
// for libm expf, and MORE accurate in effect: measured on the 246000-sample fixture, logits vs HF fp64 7.4e-5 (expf)
// -> 6.5e-5, CTC loss error 5.0e-3 -> 2.5e-3.  The bare v_exp_f32(x * log2e) is not usable in fp32: the rounding of
// the product at |x| ~ 20 is a 1e-6 relative error in every probability, and it moved that CTC loss by 2e-2.
__device__ __forceinline__ float exp_compensated(float x) {
    constexpr float L2E_HI = 1.44269504088896340736f, L2E_LO = 1.925963033500822e-08f, LN2 = 0.69314718055994530942f;
    x = fmaxf(x, -1.0e30f);                 // -inf (padding keys) would turn the residual into inf - inf
    const float y = x * L2E_HI;
    const float e = fmaf(x, L2E_LO, fmaf(x, L2E_HI, -y));
    return __builtin_amdgcn_exp2f(y) * fmaf(e, LN2, 1.0f);
}

struct AttnArgs {
    const float* qkv;           // (B, T, 3H): q | k | v
    const int32_t* frame_len;   // (B) or null
    float* ctx;                 // (B, T, H)
    int B, T, H, heads;
    float scale;
};

__device__ __forceinline__ float f4get(const float4& v, int e) {
    return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
}

typedef __attribute__((address_space(3))) float lds_f32;
typedef const __attribute__((address_space(1))) float glb_f32;
__device__ __forceinline__ void dma16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((glb_f32*)g, (lds_f32*)l, 16, 0, 0);
}

// DH = head size, NW = waves per block (32 queries each).
// K/V tiles go HBM/L2 -> LDS by LDS-DMA (1 KiB per wave instruction), double-buffered, one barrier per
// tile.  The DMA writes LDS lane-linearly, so the K image (read as 16-byte column slices by 16 rows at a
// time) is XOR-swizzled on the global SOURCE address and again on the read; V is read row-contiguously
// and needs no swizzle.
template <int DH, int NW, bool TRAIN = false>
__global__ __launch_bounds__(NW * 64) void attention_kernel(AttnArgs a, AttnTrain tr) {
    constexpr int JD = DH / 8;            // 8-wide d blocks for the QK^T contraction
    constexpr int DT = DH / 32;           // 32-wide d tiles of the output
    constexpr int SPR = DH / 4;           // 16-B slots per row
    constexpr int RPP = 256 / DH;         // rows per 1-KiB DMA piece
    constexpr int NP = KT * DH / 256;     // pieces per operand per tile
    constexpr int SH = DH == 32 ? 1 : 0;  // swizzle: slot ^= (row >> SH) & SWM
    constexpr int SWM = (SPR < 16 ? SPR : 16) - 1;
    constexpr int STAGE = 2 * KT * DH;    // floats per buffer (K then V)
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (an SGPR: the DMA's LDS address is then provably wave-uniform)
    const int li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const int64_t ld = 3 * (int64_t)a.H;
    const float* __restrict__ base = a.qkv + (int64_t)b * a.T * ld + head * DH;
    const int flen = a.frame_len ? a.frame_len[b] : a.T;

    // ---- Q fragment (B operand of S^T): lane = (query li, d half lh), pre-scaled (encoder.py:28) ----
    float4 qf[JD];
    {
        const int qr = min(q0 + li, a.T - 1);
        const float* qp = base + (int64_t)qr * ld + 4 * lh;
#pragma unroll
        for (int j = 0; j < JD; ++j) {
            float4 v = *reinterpret_cast<const float4*>(qp + 8 * j);
            qf[j] = make_float4(v.x * a.scale, v.y * a.scale, v.z * a.scale, v.w * a.scale);
        }
    }

    // ---- tile DMA: piece p (wave-uniform) covers rows p*RPP .. of K (p < NP) or V (p >= NP) ----
    const int p_row = lane / SPR, p_slot = lane % SPR;
    auto issue_tile = [&](int tile, int buf) {
        float* S = smem + buf * STAGE;
        const int k0 = tile * KT;
        // A FIXED number of pieces per wave, straight-line (pieces past the end wrap around and reload a piece another wave also
        // loads -- same bytes, harmless): with a data-dependent trip count (`p = wave; p < 2 NP; p += NW`, wave in a VGPR) the
        // loop is exec-masked control flow, after which the compiler's wait-count pass drains vmcnt(0) before the first LDS read
        // of the tile -- i.e. this block waited for the NEXT tile's K / V before computing on the current one.
        constexpr int PW = (2 * NP + NW - 1) / NW;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int p = (wave + i * NW) % (2 * NP);
            const bool isv = p >= NP;
            const int pp = isv ? p - NP : p;
            const int r = pp * RPP + p_row;                       // row inside the tile
            const int key = min(k0 + r, a.T - 1);                 // clamp: tail rows are masked / weighted 0
            const int slot = isv ? p_slot : (p_slot ^ ((r >> SH) & SWM));
            dma16(base + (int64_t)key * ld + (isv ? 2 : 1) * a.H + slot * 4, S + p * 256);
        }
    };

    f32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    // dropout hash inputs hoisted out of the tile loop: element index = ((b h + head) T + q) T + key, modulo 2^32
    const uint32_t drop_key = TRAIN ? dropout_key(tr.seed, tr.stream) : 0u, drop_thr = TRAIN ? dropout_threshold(tr.p) : 0u;
    const uint32_t drop_row = (uint32_t)((((uint64_t)b * a.heads + head) * a.T + (uint64_t)min(q0 + li, a.T - 1)) * attention_drop_stride(a.T));

    const int ntiles = (a.T + KT - 1) / KT;
    issue_tile(0, 0);
    __syncthreads();                       // carries the vmcnt(0) that retires the DMA

    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * KT, buf = tile & 1;
        // unconditional (the last iteration re-requests its own tile into the idle stage): behind a branch the compiler's wait-count
        // pass drains vmcnt(0) before the first LDS read of the tile, i.e. DMA and compute no longer overlap inside a block
        issue_tile(min(tile + 1, ntiles - 1), buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const float* Ks = smem + buf * STAGE;
        const float* Vs = Ks + KT * DH;

        // ---- S^T = K Q^T for two 32-key sub-tiles ----
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
            const int row = kt * 32 + li;
            const float* kp = Ks + row * DH;
            const int sw = (row >> SH) & SWM;
#pragma unroll
            for (int j = 0; j < JD; ++j) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + (((2 * j + lh) ^ sw) << 2));
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], f4get(qf[j], e), s[kt], 0, 0, 0);
            }
        }
        // ---- mask + online softmax (lane owns query li; keys (r&3) + 8 (r>>2) + 4 lh) ----
        // Per-score VALU work competes with the other waves' MFMA issue, so it is kept minimal: masking only on tiles
        // that touch the valid-length / T boundary (wave-uniform branch), dropout hash inputs hoisted.
        if (k0 + KT > min(flen, a.T)) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v = s[kt][r];
                    v = key >= flen ? v - 10000.0f : v;       // (1 - mask) * -10000, encoder.py:256-257
                    v = key >= a.T ? -INFINITY : v;           // tile padding: not a key at all
                    s[kt][r] = v;
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);          // exp(-inf) = 0 on the first tile
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = exp_compensated(s[kt][r] - m_new);
                s[kt][r] = p;
                rs += p;
            }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        if (TRAIN && tr.p > 0.f) {
            // attention-probability dropout (encoder.py:42-44): the row sum above uses the un-dropped p; the
            // 1 / (1 - p) factor is applied once, with the final normalisation
            const uint32_t cbase = drop_row + (uint32_t)k0;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t col = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    s[kt][r] = dropout_keep32(drop_key, cbase + attention_drop_col(col), drop_thr) ? s[kt][r] : 0.f;
                }
        }
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

        // ---- O^T += V^T P^T : A = V[key][d0 + li], B = p (already in B-operand position) ----
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* vp = Vs + (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * DH + li;
#pragma unroll
                for (int d = 0; d < DT; ++d)
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32 * d], s[kt][r], o[d], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();            // next tile landed (vmcnt 0) and every wave is done with this one
    }

    // ---- normalise and store: O^T rows are d = 32 dt + (r&3) + 8 (r>>2) + 4 lh, column = query ----
    const int q = q0 + li;
    if (TRAIN && q < a.T && lh == 0) tr.lse[((int64_t)b * a.heads + head) * a.T + q] = m_run + logf(l_run);
    if (q < a.T) {
        const float inv = ((TRAIN && tr.p > 0.f) ? 1.0f / (1.0f - tr.p) : 1.0f) / l_run;
        float* op = a.ctx + ((int64_t)b * a.T + q) * a.H + head * DH + 4 * lh;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(op + 32 * d + 8 * g) =
                    make_float4(o[d][4 * g] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv,
                                o[d][4 * g + 3] * inv);
    }
}

template <int DH, int NW>
int launch_attn_nw(const AttnArgs& a, hipStream_t s) {
    const size_t lds = (size_t)2 * 2 * KT * DH * sizeof(float);
    static std::atomic<bool> attr_set{false};   // (idempotent call; atomic so concurrent host threads agree on the flag)
    if (!attr_set) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<DH, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int qb = NW * 32;
    dim3 grid((a.T + qb - 1) / qb, a.heads, a.B), block(NW * 64);
    W2V2_LAUNCH((attention_kernel<DH, NW>), grid, block, lds, s, a, AttnTrain{0.f, 0, 0, nullptr});
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

// Pick the queries-per-block that fills 256 CUs in the fewest rounds.  Waves per block stay a multiple
// of 4 (one per SIMD; 6 waves measured 77 TF vs 95-105: two SIMDs carry double load).  Measured at
// T=768, B=32, 12 heads: 4 waves (2 blocks/CU) 95 TF, 8 waves 93, 12 waves (3 per SIMD) 105.
template <int DH>
int launch_attn(const AttnArgs& a, hipStream_t s) {
    const int forced = tune_int("W2V2_ATTN_NW", -1);
    const int cand[3] = {12, 8, 4};
    int best = 4;
    int64_t best_cost = INT64_MAX;
    for (int i = 0; i < 3; ++i) {
        const int nw = cand[i];
        if (DH == 128 && nw == 12) continue;                  // 241 VGPRs: 3 waves per SIMD would spill
        const int bpc = (DH != 128 && nw == 4) ? 2 : 1;       // blocks per CU the VGPR / LDS budget admits
        const int64_t nblk = (int64_t)((a.T + nw * 32 - 1) / (nw * 32)) * a.heads * a.B;
        const int64_t rounds = (nblk + 256 * bpc - 1) / (256 * bpc);
        const int64_t cost = rounds * nw * bpc * (nw == 12 ? 9 : 10);   // 3 waves per SIMD run ~10 % better
        if (cost < best_cost) { best_cost = cost; best = nw; }
    }
    if (forced > 0) best = forced;
    if constexpr (DH == 128) {
        return best == 8 ? launch_attn_nw<DH, 8>(a, s) : launch_attn_nw<DH, 4>(a, s);
    } else {
        switch (best) {
            case 6: return launch_attn_nw<DH, 6>(a, s);
            case 8: return launch_attn_nw<DH, 8>(a, s);
            case 12: return launch_attn_nw<DH, 12>(a, s);
            default: return launch_attn_nw<DH, 4>(a, s);
        }
    }
}


// ======================================================================================
// Training: forward with attention-probability dropout + saved log-sum-exp, and backward.
// The backward recomputes P from q, k and the saved lse (nothing T x T is ever stored):
//   dP = (dO V^T) * keep/(1-p);   dS = P * (dP - D),  D[q] = sum_d dO[q,d] O[q,d];
//   dQ = scale * dS K;   dK = scale * dS^T Q;   dV = Pd^T dO,  Pd = P * keep/(1-p).
// Both kernels reuse the forward's layout trick: the wave that owns a COLUMN (a query for dQ, a key
// for dK/dV) holds that column's [col][d] fragments in registers as MFMA B operands, the other side
// streams through LDS tiles, and every T x T quantity (S, P, dP, dS) lives only in the MFMA C/D
// registers of the lane that owns the column -- which is exactly the B-operand position of the next
// contraction.
// ======================================================================================
template <int DH>
struct Swz {
    static constexpr int SPR = DH / 4, RPP = 256 / DH, NP = KT * DH / 256;
    static constexpr int SH = DH == 32 ? 1 : 0, SWM = (SPR < 16 ? SPR : 16) - 1;
    __device__ static __forceinline__ int f(int row) { return (row >> SH) & SWM; }
};

// DMA one KT x DH tile of rows [r0, r0 + KT) (clamped to T - 1) at column offset `col` of the packed
// (B, T, ld) buffer into LDS at `dst`, XOR-swizzled; pieces striped over the block's NW waves.
template <int DH, int NW>
__device__ __forceinline__ void dma_tile_swz(const float* base, int64_t ld, int col, int r0, int T,
                                              float* dst, int wave, int lane, int piece0) {
    using Z = Swz<DH>;
    const int p_row = lane / Z::SPR, p_slot = lane % Z::SPR;
    constexpr int PW = (Z::NP + NW - 1) / NW;      // fixed, straight-line (see attention_kernel's issue_tile)
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int p = (wave + i * NW) % Z::NP;
        const int r = p * Z::RPP + p_row;
        const int row = min(r0 + r, T - 1);
        dma16(base + (int64_t)row * ld + col + ((p_slot ^ Z::f(r)) << 2), dst + (piece0 + p) * 256);
    }
}

// D[b, h, t] = sum_d dO[b, t, h, d] * O[b, t, h, d]
template <int DH>
__global__ void attn_dvec_kernel(const float* __restrict__ o, const float* __restrict__ d_o, float* __restrict__ dvec,
                                 int B, int T, int H, int heads) {
    const int64_t row = blockIdx.x;            // (b, t)
    const int b = (int)(row / T), t = (int)(row % T);
    for (int h = threadIdx.x >> 6; h < heads; h += blockDim.x >> 6) {
        const int lane = threadIdx.x & 63;
        float acc = 0.f;
        for (int d = lane; d < DH; d += 64) {
            const int64_t i = row * H + h * DH + d;
            acc += o[i] * d_o[i];
        }
        acc = wave_sum(acc);
        if (lane == 0) dvec[((int64_t)b * heads + h) * T + t] = acc;
    }
}

struct AttnBwdArgs {
    const float* qkv;
    const int32_t* frame_len;
    const float* d_o;       // (B, T, H)
    const float* dvec;      // (B, heads, T)
    float* dqkv;            // (B, T, 3H)
    int B, T, H, heads;
    float scale;
};

// ---- dQ: block = 4 waves x 32 queries; loop over 64-key tiles of K and V ----
template <int DH>
__global__ __launch_bounds__(256) void attention_bwd_dq_kernel(AttnBwdArgs a, AttnTrain tr) {
    using Z = Swz<DH>;
    constexpr int NW = 4, JD = DH / 8, DT = DH / 32, STAGE = 2 * KT * DH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const int64_t ld = 3 * (int64_t)a.H;
    const float* __restrict__ base = a.qkv + (int64_t)b * a.T * ld + head * DH;
    const int flen = a.frame_len ? a.frame_len[b] : a.T;
    const int qr = min(q0 + li, a.T - 1);
    const bool qok = q0 + li < a.T;

    float4 qf[JD], dof[JD];
    {
        const float* qp = base + (int64_t)qr * ld + 4 * lh;
        const float* dp = a.d_o + ((int64_t)b * a.T + qr) * a.H + head * DH + 4 * lh;
#pragma unroll
        for (int j = 0; j < JD; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(qp + 8 * j);
            qf[j] = make_float4(v.x * a.scale, v.y * a.scale, v.z * a.scale, v.w * a.scale);
            dof[j] = *reinterpret_cast<const float4*>(dp + 8 * j);
        }
    }
    const int64_t sidx = ((int64_t)b * a.heads + head) * a.T + qr;
    const float lse = tr.lse[sidx], dv = a.dvec[sidx];
    const float inv = tr.p > 0.f ? 1.0f / (1.0f - tr.p) : 1.0f;
    const uint64_t rowbase = (uint64_t)sidx * attention_drop_stride(a.T);
    const uint32_t drop_key = dropout_key(tr.seed, tr.stream), drop_thr = dropout_threshold(tr.p);

    f32x16 dq[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;

    auto issue = [&](int tile, int buf) {
        float* S = smem + buf * STAGE;
        dma_tile_swz<DH, NW>(base, ld, a.H, tile * KT, a.T, S, wave, lane, 0);              // K
        dma_tile_swz<DH, NW>(base, ld, 2 * a.H, tile * KT, a.T, S, wave, lane, Z::NP);      // V
    };
    const int ntiles = (a.T + KT - 1) / KT;
    issue(0, 0);
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * KT, buf = tile & 1;
        issue(min(tile + 1, ntiles - 1), buf ^ 1);      // unconditional: see attention.hip (a branch here costs the DMA / compute overlap)
        __builtin_amdgcn_sched_barrier(0);
        const float* Ks = smem + buf * STAGE;
        const float* Vs = Ks + KT * DH;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
            const int row = kt * 32 + li, sw = Z::f(row);
#pragma unroll
            for (int j = 0; j < JD; ++j) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + row * DH + (((2 * j + lh) ^ sw) << 2));
                const f32x4 vf = *reinterpret_cast<const f32x4*>(Vs + row * DH + (((2 * j + lh) ^ sw) << 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], f4get(qf[j], e), s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[e], f4get(dof[j], e), dp, 0, 0, 0);
                }
            }
            // dS^T = P^T * (dP^T * keep/(1-p) - D);  P = exp(S - lse)
            if (k0 + KT > min(flen, a.T)) {         // boundary tiles only (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float sv = s[r];
                    sv = key >= flen ? sv - 10000.0f : sv;
                    s[r] = key < a.T ? sv : -INFINITY;       // exp2(-inf) = 0: a padding key contributes nothing
                }
            }
            const uint32_t cbase = (uint32_t)rowbase + (uint32_t)(k0 + kt * 32 + 8 * lh);      // attention_drop_col of the lane's keys: 16 a + 8 lh + 4 b + c
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = exp_compensated(s[r] - lse);
                float g = dp[r];
                if (tr.p > 0.f) g = dropout_keep32(drop_key, cbase + (uint32_t)((r & 3) + 4 * ((r >> 2) & 1) + 16 * (r >> 3)), drop_thr) ? g : 0.f;
                s[r] = pv * fmaf(g, inv, -dv);
            }
            // dQ^T[d][q] += sum_key K[key][d] dS^T[key][q]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int ksw = Z::f(krow);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const int dd = 32 * d + li;
                    const float kv = Ks[krow * DH + ((((dd >> 2) ^ ksw)) << 2) + (dd & 3)];
                    dq[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, s[r], dq[d], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    if (qok) {
        float* op = a.dqkv + ((int64_t)b * a.T + q0 + li) * ld + head * DH + 4 * lh;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(op + 32 * d + 8 * g) =
                    make_float4(dq[d][4 * g] * a.scale, dq[d][4 * g + 1] * a.scale, dq[d][4 * g + 2] * a.scale,
                                dq[d][4 * g + 3] * a.scale);
    }
}

// ---- dK, dV: block = 4 waves x 32 keys; loop over 64-query tiles of Q and dO (+ their lse, D) ----
template <int DH>
__global__ __launch_bounds__(256) void attention_bwd_dkv_kernel(AttnBwdArgs a, AttnTrain tr) {
    using Z = Swz<DH>;
    constexpr int NW = 4, JD = DH / 8, DT = DH / 32, STAGE = 2 * KT * DH + 2 * KT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int c0 = (blockIdx.x * NW + wave) * 32;          // this wave's 32 keys
    const int64_t ld = 3 * (int64_t)a.H;
    const float* __restrict__ base = a.qkv + (int64_t)b * a.T * ld + head * DH;
    const float* __restrict__ dobase = a.d_o + (int64_t)b * a.T * a.H + head * DH;
    const int flen = a.frame_len ? a.frame_len[b] : a.T;
    const int key = c0 + li;
    const int kr = min(key, a.T - 1);
    const bool kok = key < a.T;
    const float kmask = key >= flen ? -10000.0f : 0.0f;
    const uint32_t drop_key = dropout_key(tr.seed, tr.stream), drop_thr = dropout_threshold(tr.p);

    float4 kf[JD], vf[JD];
    {
        const float* kp = base + (int64_t)kr * ld + a.H + 4 * lh;
        const float* vp = base + (int64_t)kr * ld + 2 * a.H + 4 * lh;
#pragma unroll
        for (int j = 0; j < JD; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(kp + 8 * j);
            kf[j] = make_float4(v.x * a.scale, v.y * a.scale, v.z * a.scale, v.w * a.scale);   // S = scale q.k
            vf[j] = *reinterpret_cast<const float4*>(vp + 8 * j);
        }
    }
    const float inv = tr.p > 0.f ? 1.0f / (1.0f - tr.p) : 1.0f;
    const int64_t bh = (int64_t)b * a.heads + head;
    const uint32_t drop_col = (uint32_t)((uint64_t)bh * a.T * attention_drop_stride(a.T)) + attention_drop_col((uint32_t)kr);

    f32x16 dk[DT], dvv[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[d][r] = dvv[d][r] = 0.f;

    auto issue = [&](int tile, int buf) {
        float* S = smem + buf * STAGE;
        dma_tile_swz<DH, NW>(base, ld, 0, tile * KT, a.T, S, wave, lane, 0);                    // Q
        dma_tile_swz<DH, NW>(dobase, a.H, 0, tile * KT, a.T, S, wave, lane, Z::NP);             // dO
        if (tid < 2 * KT) {      // lse and D of the tile's queries (plain stores: visible after the barrier)
            const int qq = min(tile * KT + (tid & (KT - 1)), a.T - 1);
            S[2 * KT * DH + tid] = tid < KT ? tr.lse[bh * a.T + qq] : a.dvec[bh * a.T + qq];
        }
    };
    const int ntiles = (a.T + KT - 1) / KT;
    issue(0, 0);
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int t0 = tile * KT, buf = tile & 1;
        issue(min(tile + 1, ntiles - 1), buf ^ 1);      // unconditional: see attention.hip (a branch here costs the DMA / compute overlap)
        __builtin_amdgcn_sched_barrier(0);
        const float* Qs = smem + buf * STAGE;
        const float* Os = Qs + KT * DH;
        const float* Ls = Os + KT * DH;      // [0, KT): lse, [KT, 2KT): D
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
            const int row = qt * 32 + li, sw = Z::f(row);
#pragma unroll
            for (int j = 0; j < JD; ++j) {
                const f32x4 qv = *reinterpret_cast<const f32x4*>(Qs + row * DH + (((2 * j + lh) ^ sw) << 2));
                const f32x4 ov = *reinterpret_cast<const f32x4*>(Os + row * DH + (((2 * j + lh) ^ sw) << 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[e], f4get(kf[j], e), s, 0, 0, 0);     // S[q][key]
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(ov[e], f4get(vf[j], e), dp, 0, 0, 0);   // dP[q][key]
                }
            }
            // lane owns key column `key`; register r is query row qt*32 + (r&3) + 8 (r>>2) + 4 lh
            // (columns of keys >= T are clamped duplicates whose results are never stored, so only the QUERY bound
            // needs masking, and only on the last tile)
            const bool qtail = t0 + KT > a.T;
            const uint32_t didx = drop_col + (uint32_t)(t0 + qt * 32 + 4 * lh) * attention_drop_stride(a.T);   // ((bh T + q) T + key) mod 2^32
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float pv = exp_compensated(s[r] + kmask - Ls[ql]);
                if (qtail) pv = t0 + ql < a.T ? pv : 0.f;
                float g = dp[r], pd = pv;
                if (tr.p > 0.f) {
                    const bool keep = dropout_keep32(drop_key, didx + (uint32_t)((r & 3) + 8 * (r >> 2)) * attention_drop_stride(a.T), drop_thr);
                    g = keep ? g : 0.f;
                    pd = keep ? pv * inv : 0.f;
                }
                dp[r] = pd;                                    // Pd[q][key]
                s[r] = pv * fmaf(g, inv, -Ls[KT + ql]);        // dS[q][key]
            }
            // dV^T[d][key] += sum_q dO[q][d] Pd[q][key];   dK^T[d][key] += sum_q Q[q][d] dS[q][key]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qrow = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int qsw = Z::f(qrow);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const int dd = 32 * d + li;
                    const int off = qrow * DH + (((dd >> 2) ^ qsw) << 2) + (dd & 3);
                    dvv[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(Os[off], dp[r], dvv[d], 0, 0, 0);
                    dk[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[off], s[r], dk[d], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    if (kok) {
        float* kp = a.dqkv + ((int64_t)b * a.T + key) * ld + a.H + head * DH + 4 * lh;
        float* vp = kp + a.H;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // kf was pre-scaled for S, so dS is the gradient of the SCALED score: dK = scale * dS^T Q
                *reinterpret_cast<float4*>(kp + 32 * d + 8 * g) =
                    make_float4(dk[d][4 * g] * a.scale, dk[d][4 * g + 1] * a.scale, dk[d][4 * g + 2] * a.scale, dk[d][4 * g + 3] * a.scale);
                *reinterpret_cast<float4*>(vp + 32 * d + 8 * g) =
                    make_float4(dvv[d][4 * g], dvv[d][4 * g + 1], dvv[d][4 * g + 2], dvv[d][4 * g + 3]);
            }
    }
}

template <int DH, int NW>
int launch_attn_train_impl(const AttnArgs& a, const AttnTrain& tr, hipStream_t s) {
    const size_t lds = (size_t)2 * 2 * KT * DH * sizeof(float);
    static std::atomic<bool> attr_set{false};   // (idempotent call; atomic so concurrent host threads agree on the flag)
    if (!attr_set) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<DH, NW, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid((a.T + NW * 32 - 1) / (NW * 32), a.heads, a.B), block(NW * 64);
    W2V2_LAUNCH((attention_kernel<DH, NW, true>), grid, block, lds, s, a, tr);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

// 12 waves per block (3 per SIMD) when the sequence is long enough to fill whole blocks, as the inference launcher picks
template <int DH>
int launch_attn_train(const AttnArgs& a, const AttnTrain& tr, hipStream_t s) {
    if constexpr (DH == 64) {
        if (a.T >= 384) return launch_attn_train_impl<DH, 12>(a, tr, s);
    }
    return launch_attn_train_impl<DH, 4>(a, tr, s);
}

template <int DH>
int launch_attn_bwd(const AttnBwdArgs& a, const AttnTrain& tr, const float* ctx, float* dvec, hipStream_t s) {
    W2V2_LAUNCH(attn_dvec_kernel<DH>, dim3((unsigned)((int64_t)a.B * a.T)), dim3(256), 0, s, ctx, a.d_o, dvec,
                       a.B, a.T, a.H, a.heads);
    const size_t lds_q = (size_t)2 * 2 * KT * DH * sizeof(float);
    const size_t lds_kv = (size_t)2 * (2 * KT * DH + 2 * KT) * sizeof(float);
    W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_dq_kernel<DH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q));
    W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_dkv_kernel<DH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv));
    dim3 grid((a.T + 127) / 128, a.heads, a.B), block(256);
    W2V2_LAUNCH(attention_bwd_dq_kernel<DH>, grid, block, lds_q, s, a, tr);
    W2V2_LAUNCH(attention_bwd_dkv_kernel<DH>, grid, block, lds_kv, s, a, tr);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace

int launch_attention(Profiler* prof, const float* qkv, const int32_t* frame_len, float* ctx, int B,
                     int T, int H, int heads, hipStream_t s) {
    return launch_attention_x(prof, qkv, nullptr, frame_len, ctx, B, T, H, heads, nullptr, s);
}

int launch_attention_x(Profiler* prof, const float* qkv, const uint16_t* qkv16, const int32_t* frame_len, float* ctx, int B, int T, int H,
                       int heads, uint16_t* ctx16, hipStream_t s, const PlaneOut* planes) {
    W2V2_REQUIRE(B > 0 && T > 0 && heads > 0 && H % heads == 0, "attention: bad sizes");
    const int dh = H / heads;
    ProfScope ps(prof, FAM_ATTENTION, 4.0 * B * (double)heads * T * (double)T * dh,
                 4.0 * B * (double)T * 4.0 * H, s);
    if (gemm_get_precision() == 1 && attention_bf16_supported(dh))
        return launch_attention_fwd_bf16(qkv, qkv16, frame_len, ctx, ctx16, B, T, H, heads, nullptr, s);
    W2V2_REQUIRE(qkv && (ctx || (planes && planes->p)), "attention: null operand");
    W2V2_REQUIRE(!ctx16, "attention: a bf16 shadow output needs the bf16 kernel (precision 1, head size 64)");
    if (gemm_get_precision() >= 2 && tune_int("W2V2_SPLIT_ATTN", 1) != 0 && attention_split_supported(dh) && H % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(ctx)) & 15) == 0)
        return launch_attention_split(qkv, frame_len, ctx, B, T, H, heads, s, planes, gemm_get_precision() == W2V2_PRECISION_F16X2 ? PF_F16X2 : PF_BF16X3,
                                      planes ? planes->range_flag : nullptr);     // fp32-level results, bf16 / fp16 matrix cores
    W2V2_REQUIRE(ctx && !(planes && planes->p), "attention: a plane output needs the split kernel (precision modes 2 / 3, head size 64)");
    AttnArgs a{qkv, frame_len, ctx, B, T, H, heads, 1.0f / sqrtf((float)dh)};
    switch (dh) {
        case 32: return launch_attn<32>(a, s);
        case 64: return launch_attn<64>(a, s);
        case 128: return launch_attn<128>(a, s);
        default:
            set_error("attention: head size %d unsupported (32, 64, 128)", dh);
            return W2V2_EINVAL;
    }
}

}  // namespace w2v2

namespace w2v2 {

int launch_attention_train(Profiler* prof, const float* qkv, const int32_t* frame_len, float* ctx, int B, int T,
                           int H, int heads, const AttnTrain& tr, hipStream_t s) {
    return launch_attention_train_x(prof, qkv, nullptr, frame_len, ctx, nullptr, B, T, H, heads, tr, s);
}

int launch_attention_train_x(Profiler* prof, const float* qkv, const uint16_t* qkv16, const int32_t* frame_len, float* ctx, uint16_t* ctx16,
                             int B, int T, int H, int heads, const AttnTrain& tr, hipStream_t s) {
    W2V2_REQUIRE((qkv || qkv16) && (ctx || ctx16) && tr.lse, "attention_train: null operand");
    W2V2_REQUIRE(B > 0 && T > 0 && heads > 0 && H % heads == 0 && tr.p >= 0.f && tr.p < 1.f, "attention_train: bad sizes");
    const int dh = H / heads;
    ProfScope ps(prof, FAM_ATTENTION, 4.0 * B * (double)heads * T * (double)T * dh, 4.0 * B * (double)T * 4.0 * H, s);
    if (gemm_get_precision() == 1 && attention_bf16_supported(dh))
        return launch_attention_fwd_bf16(qkv, qkv16, frame_len, ctx, ctx16, B, T, H, heads, &tr, s);
    W2V2_REQUIRE(qkv && ctx, "attention_train: the fp32 kernels need the fp32 qkv and write the fp32 ctx");
    AttnArgs a{qkv, frame_len, ctx, B, T, H, heads, 1.0f / sqrtf((float)dh)};
    W2V2_REQUIRE(!ctx16, "attention_train: a bf16 shadow output needs the bf16 kernel (precision 1, head size 64)");
    switch (dh) {
        case 32: return launch_attn_train<32>(a, tr, s);
        case 64: return launch_attn_train<64>(a, tr, s);
        default:
            set_error("attention_train: head size %d unsupported (32, 64)", dh);
            return W2V2_EINVAL;
    }
}

int launch_attention_bwd(Profiler* prof, const float* qkv, const int32_t* frame_len, const float* ctx,
                         const float* dctx, float* dqkv, float* dvec_ws, int B, int T, int H, int heads,
                         const AttnTrain& tr, hipStream_t s, uint16_t* dqkv16, const uint16_t* qkv16, const uint16_t* dctx16, float* colpart,
                         const uint16_t* ctx16) {
    W2V2_REQUIRE((qkv || qkv16) && (ctx || ctx16) && (dctx || dctx16) && (dqkv || dqkv16) && dvec_ws && tr.lse, "attention_bwd: null operand");
    W2V2_REQUIRE(B > 0 && T > 0 && heads > 0 && H % heads == 0, "attention_bwd: bad sizes");
    const int dh = H / heads;
    ProfScope ps(prof, FAM_ATTENTION, 10.0 * B * (double)heads * T * (double)T * dh, 4.0 * B * (double)T * 8.0 * H, s);
    if (gemm_get_precision() == 1 && attention_bf16_supported(dh)) {
        // (D = rowsum(dO o O) is computed by the dQ kernel from the bf16 values of dO and O, whichever form they arrive in)
        return launch_attention_bwd_bf16(qkv, qkv16, frame_len, dctx, dctx16, dvec_ws, dqkv, dqkv16, B, T, H, heads, tr, s, colpart, ctx, ctx16);
    }
    W2V2_REQUIRE(qkv && dqkv && ctx && dctx && !colpart, "attention_bwd: the fp32 kernels need the fp32 qkv / ctx / dctx / dqkv and leave no column sums");
    AttnBwdArgs a{qkv, frame_len, dctx, dvec_ws, dqkv, B, T, H, heads, 1.0f / sqrtf((float)dh)};
    W2V2_REQUIRE(!dqkv16, "attention_bwd: a bf16 shadow of dqkv is only written by the bf16 kernels (head size 64, precision mode 1)");
    switch (dh) {
        case 32: return launch_attn_bwd<32>(a, tr, ctx, dvec_ws, s);
        case 64: return launch_attn_bwd<64>(a, tr, ctx, dvec_ws, s);
        default:
            set_error("attention_bwd: head size %d unsupported (32, 64)", dh);
            return W2V2_EINVAL;
    }
}

}  // namespace w2v2
