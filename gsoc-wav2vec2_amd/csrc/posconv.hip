// Relative positional convolution embedding, fused:
//   y = xz + GELU( grouped_conv1d(pad(xz, K/2, K/2), W_eff)[:T] + bias )
// where xz is x with frames >= frame_len[b] zeroed and W_eff is the
// weight-normalised kernel.
//
// Reference: Conv1DWithWeightNorm (tensorflow_addons.py:5-58: kernel =
// l2_normalize(weight_v, axes [1,2]) * weight_g, i.e. the norm is PER KERNEL TAP;
// explicit tf.pad then a valid grouped Conv1D), PositionalConvEmbedding
// (encoder.py:153-181: pad = K // 2, drop the last frame when K is even, exact
// GELU) and the encoder's use of it (encoder.py:253,265: zero padded frames,
// then batch + pos_conv(batch)).
//
// Compute shape: per group an implicit GEMM  M = T, N = C_out/groups (48 | 64),
// K-dim = K_taps * C_in/groups (6144 | 8192).  N is not a multiple of 32, so the
// matrix core form is v_mfma_f32_16x16x4_f32 (same fp32 peak as 32x32x2).
// A block owns 128 output frames of one (batch, group): the input slab
// (128 + K - 1 frames x C_in/groups) sits in LDS once and every tap reads it at a
// shifted row -- the Toeplitz structure means no im2col and no re-fetch; the
// per-tap weight tile (C_in/g x C_out/g) is double-buffered through LDS.
#include "common.h"

namespace w2v2 {

using f32x4 = __attribute__((ext_vector_type(4))) float;

namespace {

constexpr int PBM = 128;   // output frames per block (4 waves x 32)

// ---- weight-norm + regroup -------------------------------------------------
// weight_v (K, cg, H), weight_g (K) -> wg (groups, K, cg, og); one block per tap.
__global__ __launch_bounds__(256) void weight_norm_regroup_kernel(const float* __restrict__ wv,
                                                                  const float* __restrict__ wgain,
                                                                  float* __restrict__ out, int K,
                                                                  int cg, int H, int groups) {
    __shared__ double red[4];
    const int k = blockIdx.x;
    const int n = cg * H;
    const float* v = wv + (int64_t)k * n;
    double ss = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) ss += (double)v[i] * (double)v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    ss = red[0] + red[1] + red[2] + red[3];
    // tf.nn.l2_normalize: x * rsqrt(max(sum(x^2), 1e-12)); then * weight_g
    const float scale = (float)((double)wgain[k] / sqrt(ss > 1e-12 ? ss : 1e-12));
    const int og = H / groups;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int ci = i / H, c = i % H;
        const int g = c / og, co = c % og;
        out[(((int64_t)g * K + k) * cg + ci) * og + co] = v[i] * scale;
    }
}

struct PosArgs {
    const float* x;
    const float* wg;
    const float* bias;
    const int32_t* frame_len;
    float* y;
    int B, T, H, K, groups, act;
};

// CG = channels per group (input == output), a multiple of 16, <= 64
template <int CG>
__global__ __launch_bounds__(256) void pos_conv_kernel(PosArgs a) {
    constexpr int NT = CG / 16;                 // 16-wide output column tiles per wave
    constexpr int KS = CG / 4;                  // 4-deep k steps per tap
    constexpr int XS = CG + 2;                  // slab row stride: conflict-free A-fragment b32 reads
    constexpr int WS = (CG % 32 == 0) ? CG + 16 : CG;   // weight row stride (bank offset 16 per k row)
    constexpr int WV = (CG * CG / 4 + 255) / 256;       // float4 per thread per tap tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 15, lk = lane >> 4;
    const int t0 = blockIdx.x * PBM, g = blockIdx.y, b = blockIdx.z;
    const int pad = a.K / 2;
    const int rows = PBM + a.K - 1;
    float* Xs = smem;                           // rows x XS
    float* Ws = smem + ((rows * XS + 3) & ~3);  // 2 x CG x WS
    const int flen = a.frame_len ? a.frame_len[b] : a.T;

    // ---- stage the input slab: frames t0-pad .. t0+PBM+K-2-pad, zero outside [0, min(T, flen)) ----
    const float* __restrict__ xb = a.x + (int64_t)b * a.T * a.H + g * CG;
    for (int i = tid; i < rows * (CG / 4); i += 256) {
        const int r = i / (CG / 4), c4 = (i % (CG / 4)) * 4;
        const int t = t0 - pad + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < a.T && t < flen) v = *reinterpret_cast<const float4*>(xb + (int64_t)t * a.H + c4);
        float* d = Xs + r * XS + c4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    // ---- weight tap tiles: global (g, k, ci, co) contiguous CG*CG floats per tap ----
    const float* __restrict__ wbase = a.wg + (int64_t)g * a.K * CG * CG;
    float4 wr[WV];
    auto w_load = [&](int tap) {
#pragma unroll
        for (int j = 0; j < WV; ++j) {
            const int idx = tid + 256 * j;
            wr[j] = idx < CG * CG / 4 ? *reinterpret_cast<const float4*>(wbase + (int64_t)tap * CG * CG + idx * 4)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto w_store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < WV; ++j) {
            const int idx = tid + 256 * j;
            if (idx < CG * CG / 4) {
                const int ci = (idx * 4) / CG, co = (idx * 4) % CG;
                *reinterpret_cast<float4*>(Ws + buf * CG * WS + ci * WS + co) = wr[j];
            }
        }
    };
    w_load(0);
    w_store(0);
    __syncthreads();

    f32x4 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* Xw = Xs + (wave * 32 + ln) * XS + lk;   // A: row = frame (+tap), k = channel
    for (int tap = 0; tap < a.K; ++tap) {
        const int cur = tap & 1;
        w_load(tap + 1 < a.K ? tap + 1 : tap);            // unconditional: keeps wr[] in registers
        const float* Wc = Ws + cur * CG * WS + lk * WS + ln;
        const float* Xt = Xw + tap * XS;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float a0 = Xt[ks * 4];
            const float a1 = Xt[16 * XS + ks * 4];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float bv = Wc[(ks * 4) * WS + n * 16];
                acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[0][n], 0, 0, 0);
                acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[1][n], 0, 0, 0);
            }
        }
        w_store(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: + bias -> GELU -> + xz (residual from the slab) -> store ----
    // 16x16 C/D map: col = lane & 15, row = 4 (lane >> 4) + reg
    float* __restrict__ yb = a.y + (int64_t)b * a.T * a.H + g * CG;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n * 16 + ln;
        const float bv = a.bias[g * CG + co];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = wave * 32 + m * 16 + lk * 4 + r;
                const int t = t0 + lr;
                if (t < a.T) {
                    const float res = Xs[(lr + pad) * XS + co];
                    yb[(int64_t)t * a.H + co] = res + apply_act(acc[m][n][r] + bv, a.act);
                }
            }
        }
    }
}

template <int CG>
int launch_pos(const PosArgs& a, hipStream_t s) {
    constexpr int XS = CG + 2;
    constexpr int WS = (CG % 32 == 0) ? CG + 16 : CG;
    const int rows = PBM + a.K - 1;
    const size_t lds = (size_t)(((rows * XS + 3) & ~3) + 2 * CG * WS) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pos_conv_kernel<CG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    W2V2_REQUIRE(lds <= 160 * 1024, "pos_conv: K=%d needs %zu B of LDS (> 160 KiB)", a.K, lds);
    dim3 grid((a.T + PBM - 1) / PBM, a.groups, a.B), block(256);
    hipLaunchKernelGGL(pos_conv_kernel<CG>, grid, block, lds, s, a);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace

int launch_weight_norm_regroup(Profiler* prof, const float* wv, const float* wg, float* out, int K,
                               int cg, int H, int groups, hipStream_t s) {
    W2V2_REQUIRE(wv && wg && out, "weight_norm: null operand");
    W2V2_REQUIRE(K > 0 && groups > 0 && H % groups == 0 && cg == H / groups,
                 "weight_norm: bad shape K=%d cg=%d H=%d groups=%d", K, cg, H, groups);
    ProfScope ps(prof, FAM_MISC, 3.0 * K * cg * H, 8.0 * K * cg * H, s);
    hipLaunchKernelGGL(weight_norm_regroup_kernel, dim3(K), dim3(256), 0, s, wv, wg, out, K, cg, H, groups);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_pos_conv(Profiler* prof, const float* x, const float* wg, const float* bias,
                    const int32_t* frame_len, float* y, int B, int T, int H, int K, int groups,
                    int act, hipStream_t s) {
    W2V2_REQUIRE(x && wg && bias && y, "pos_conv: null operand");
    W2V2_REQUIRE(B > 0 && T > 0 && K > 0 && groups > 0 && H % groups == 0, "pos_conv: bad sizes");
    const int cg = H / groups;
    W2V2_REQUIRE(H % 4 == 0, "pos_conv: hidden size must be a multiple of 4");
    PosArgs a{x, wg, bias, frame_len, y, B, T, H, K, groups, act};
    ProfScope ps(prof, FAM_POSCONV, 2.0 * B * (double)T * H * cg * K, 8.0 * B * (double)T * H + 4.0 * K * cg * H, s);
    switch (cg) {
        case 16: return launch_pos<16>(a, s);
        case 32: return launch_pos<32>(a, s);
        case 48: return launch_pos<48>(a, s);
        case 64: return launch_pos<64>(a, s);
        default:
            set_error("pos_conv: channels per group = %d unsupported (16, 32, 48, 64)", cg);
            return W2V2_EINVAL;
    }
}

}  // namespace w2v2
