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

namespace w2v2 {

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

constexpr int KT = 64;    // keys per tile

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
template <int DH, int NW>
__global__ __launch_bounds__(NW * 64) void attention_kernel(AttnArgs a) {
    constexpr int JD = DH / 8;            // 8-wide d blocks for the QK^T contraction
    constexpr int DT = DH / 32;           // 32-wide d tiles of the output
    constexpr int SPR = DH / 4;           // 16-B slots per row
    constexpr int RPP = 256 / DH;         // rows per 1-KiB DMA piece
    constexpr int NP = KT * DH / 256;     // pieces per operand per tile
    constexpr int SH = DH == 32 ? 1 : 0;  // swizzle: slot ^= (row >> SH) & SWM
    constexpr int SWM = (SPR < 16 ? SPR : 16) - 1;
    constexpr int STAGE = 2 * KT * DH;    // floats per buffer (K then V)
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
        for (int p = wave; p < 2 * NP; p += NW) {
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

    const int ntiles = (a.T + KT - 1) / KT;
    issue_tile(0, 0);
    __syncthreads();                       // carries the vmcnt(0) that retires the DMA

    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * KT, buf = tile & 1;
        if (tile + 1 < ntiles) issue_tile(tile + 1, buf ^ 1);     // wave-uniform branch; no registers involved
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
                const float4 kf = *reinterpret_cast<const float4*>(kp + (((2 * j + lh) ^ sw) << 2));
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4get(kf, e), f4get(qf[j], e), s[kt], 0, 0, 0);
            }
        }
        // ---- mask + online softmax (lane owns query li; keys (r&3) + 8 (r>>2) + 4 lh) ----
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = s[kt][r];
                v = key >= flen ? v - 10000.0f : v;       // (1 - mask) * -10000, encoder.py:256-257
                v = key >= a.T ? -INFINITY : v;           // tile padding: not a key at all
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);          // exp(-inf) = 0 on the first tile
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = expf(s[kt][r] - m_new);
                s[kt][r] = p;
                rs += p;
            }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
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
    if (q < a.T) {
        const float inv = 1.0f / l_run;
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
    static bool attr_set = false;
    if (!attr_set) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<DH, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int qb = NW * 32;
    dim3 grid((a.T + qb - 1) / qb, a.heads, a.B), block(NW * 64);
    hipLaunchKernelGGL((attention_kernel<DH, NW>), grid, block, lds, s, a);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

// Pick the queries-per-block that fills 256 CUs in the fewest rounds.  Waves per block stay a multiple
// of 4 (one per SIMD; 6 waves measured 77 TF vs 95-105: two SIMDs carry double load).  Measured at
// T=768, B=32, 12 heads: 4 waves (2 blocks/CU) 95 TF, 8 waves 93, 12 waves (3 per SIMD) 105.
template <int DH>
int launch_attn(const AttnArgs& a, hipStream_t s) {
    static int forced = -2;
    if (forced == -2) {
        const char* e = getenv("W2V2_ATTN_NW");     // tuning knob, not part of the ABI
        forced = e ? atoi(e) : -1;
    }
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

}  // namespace

int launch_attention(Profiler* prof, const float* qkv, const int32_t* frame_len, float* ctx, int B,
                     int T, int H, int heads, hipStream_t s) {
    W2V2_REQUIRE(qkv && ctx, "attention: null operand");
    W2V2_REQUIRE(B > 0 && T > 0 && heads > 0 && H % heads == 0, "attention: bad sizes");
    const int dh = H / heads;
    AttnArgs a{qkv, frame_len, ctx, B, T, H, heads, 1.0f / sqrtf((float)dh)};
    ProfScope ps(prof, FAM_ATTENTION, 4.0 * B * (double)heads * T * (double)T * dh,
                 4.0 * B * (double)T * 4.0 * H, s);
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
