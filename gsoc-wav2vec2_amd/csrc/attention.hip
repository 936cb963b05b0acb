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
// Block = 4 waves x 32 queries of one (batch, head); K and V tiles of 64 keys
// are staged in LDS (register-prefetched one tile ahead) and shared by the waves.
#include "common.h"

namespace w2v2 {

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

constexpr int QB = 128;   // queries per block
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

template <int DH>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
    constexpr int JD = DH / 8;          // 8-wide d blocks for the QK^T contraction
    constexpr int DT = DH / 32;         // 32-wide d tiles of the output
    constexpr int KS = DH + 4;          // K tile row stride: conflict-free ds_read_b128 column slices
    constexpr int F4 = DH / 4;          // float4 per row
    constexpr int NLD = KT * F4 / 256;  // float4 per thread per tile (K and V each)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                   // KT x KS
    float* Vs = smem + KT * KS;         // KT x DH

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * QB + wave * 32;
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

    // ---- K/V tile staging: thread -> rows (tid / F4) + (256 / F4) i, float4 column tid % F4 ----
    float4 kr[NLD], vr[NLD];
    const int s_row = tid / F4, s_c4 = (tid % F4) * 4;
    auto tile_load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int key = k0 + s_row + (256 / F4) * i;
            const bool ok = key < a.T;
            const float* p = base + (int64_t)(ok ? key : a.T - 1) * ld + s_c4;
            const float4 kv = *reinterpret_cast<const float4*>(p + a.H);
            const float4 vv = *reinterpret_cast<const float4*>(p + 2 * a.H);
            const float z = ok ? 1.0f : 0.0f;     // rows past T contribute exact zeros (never NaN * 0)
            kr[i] = make_float4(kv.x * z, kv.y * z, kv.z * z, kv.w * z);
            vr[i] = make_float4(vv.x * z, vv.y * z, vv.z * z, vv.w * z);
        }
    };
    auto tile_store = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int r = s_row + (256 / F4) * i;
            *reinterpret_cast<float4*>(Ks + r * KS + s_c4) = kr[i];
            *reinterpret_cast<float4*>(Vs + r * DH + s_c4) = vr[i];
        }
    };

    f32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (a.T + KT - 1) / KT;
    tile_load(0);
    tile_store();
    __syncthreads();

    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * KT;
        tile_load(min(tile + 1, ntiles - 1) * KT);        // unconditional prefetch (last one redundant)

        // ---- S^T = K Q^T for two 32-key sub-tiles ----
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
            const float* kp = Ks + (kt * 32 + li) * KS + 4 * lh;
#pragma unroll
            for (int j = 0; j < JD; ++j) {
                const float4 kf = *reinterpret_cast<const float4*>(kp + 8 * j);
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
        const float alpha = expf(m_run - m_new);        // exp(-inf) = 0 on the first tile
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

        __syncthreads();            // every wave is done reading this tile
        tile_store();
        __syncthreads();
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

template <int DH>
int launch_attn(const AttnArgs& a, hipStream_t s) {
    const size_t lds = (size_t)(KT * (DH + 4) + KT * DH) * sizeof(float);
    dim3 grid((a.T + QB - 1) / QB, a.heads, a.B), block(256);
    hipLaunchKernelGGL(attention_kernel<DH>, grid, block, lds, s, a);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
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
