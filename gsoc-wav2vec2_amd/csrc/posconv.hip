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
#include "gemm_epilogue.h"
#include "train.h"

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
    const float* bias;          // may be null (backward data pass)
    const int32_t* frame_len;
    float* y;
    float* pre_act;             // optional: conv + bias before the activation (saved for backward)
    int B, T, H, K, groups, act;
    int pad_left;               // K / 2 forward; K - 1 - K / 2 for the transposed (data-gradient) pass
    int add_residual;           // 1: y = xz + act(conv + bias) (forward); 0: y = act(conv + bias)
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
    // Block -> (frame block, group, sample), XCD-aware (round 5): hardware hands consecutive block ids to the eight XCDs in turn, each
    // with its own 4 MiB L2.  A block streams its group's 128 tap tiles (K cg^2 floats: 1.18 MB at cg = 48, 2.1 MB at cg = 64) from L2;
    // with the groups spread over all XCDs every L2 saw all 16 groups' 19 - 34 MB and the taps came from the Infinity Cache again and
    // again (FETCH_SIZE 1.50 GB per launch for 95 MB of operands).  Here XCD x owns groups [x groups / 8, (x + 1) groups / 8): two
    // groups' taps stay resident in its L2, and the frame blocks of one (sample, group) -- which share half their input slab --
    // follow each other on it.
    int t0, g, b;
    {
        const int tb = (a.T + PBM - 1) / PBM;
        const int bid = blockIdx.x;
        if (a.groups % 8 == 0) {
            const int gpx = a.groups >> 3, xcd = bid & 7, idx = bid >> 3;
            const int tblk = idx % tb, r = idx / tb;
            t0 = tblk * PBM;
            g = xcd * gpx + r % gpx;
            b = r / gpx;
        } else {
            t0 = (bid % tb) * PBM;
            g = (bid / tb) % a.groups;
            b = bid / (tb * a.groups);
        }
    }
    const int pad = a.pad_left;
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
        const float bv = a.bias ? a.bias[g * CG + co] : 0.0f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = wave * 32 + m * 16 + lk * 4 + r;
                const int t = t0 + lr;
                if (t < a.T) {
                    const float res = a.add_residual ? Xs[(lr + pad) * XS + co] : 0.0f;
                    const float c = acc[m][n][r] + bv;
                    if (a.pre_act) a.pre_act[((int64_t)b * a.T + t) * a.H + g * CG + co] = c;
                    yb[(int64_t)t * a.H + co] = res + apply_act(c, a.act);
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
    static std::atomic<bool> attr_set{false};   // (idempotent call; atomic so concurrent host threads agree on the flag)
    if (!attr_set) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pos_conv_kernel<CG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    W2V2_REQUIRE(lds <= 160 * 1024, "pos_conv: K=%d needs %zu B of LDS (> 160 KiB)", a.K, lds);
    dim3 grid((unsigned)(((a.T + PBM - 1) / PBM) * a.groups * a.B)), block(256);
    W2V2_LAUNCH(pos_conv_kernel<CG>, grid, block, lds, s, a);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}


// ---- training: transposed weights for the data-gradient pass ---------------------------------
// wg (g, k, ci, co) -> wg_t (g, K-1-k, co, ci):  dxz = conv(dc, wg_t) with pad_left = K - 1 - K/2
__global__ void flip_regroup_kernel(const float* __restrict__ wg, float* __restrict__ wt, int K, int cg, int groups) {
    const int64_t n = (int64_t)groups * K * cg * cg;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int co = (int)(i % cg), ci = (int)((i / cg) % cg), k = (int)((i / ((int64_t)cg * cg)) % K), g = (int)(i / ((int64_t)cg * cg * K));
        wt[(((int64_t)g * K + (K - 1 - k)) * cg + co) * cg + ci] = wg[i];
    }
}

// ---- training: weight gradient  dWg[g][k][ci][co] = sum_{b,t} xz[b][t + k - P][g cg + ci] dc[b][t][g cg + co]
// One block per (8-tap group, conv group): it walks the whole batch and all frames, so the reduction
// over (b, t) stays in MFMA accumulators (no partial slabs, no atomics).  v_mfma_f32_16x16x4_f32 with
// M = ci, N = co and the contraction over time: A[i = ci][k = t] and B[k = t][j = co] are both read
// from LDS tiles of the input slab / the upstream gradient.
constexpr int DW_TT = 128;     // frames per staged tile
constexpr int DW_TAPS = 8;     // taps per block (2 per wave)

struct PosDwArgs {
    const float* xz;   // (B, T, H) input of the conv (already masked)
    const float* dc;   // (B, T, H) gradient w.r.t. conv + bias
    float* dwg;        // (groups, K, cg, cg)
    int B, T, H, K, groups, pad;
};

template <int CG>
__global__ __launch_bounds__(256) void pos_conv_dw_kernel(PosDwArgs a) {
    constexpr int NT = CG / 16;
    constexpr int XS = CG + 2;
    constexpr int XROWS = DW_TT + DW_TAPS - 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                               // XROWS x XS
    float* Ds = smem + ((XROWS * XS + 3) & ~3);     // DW_TT x XS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 15, lk = lane >> 4;
    const int k0 = blockIdx.x * DW_TAPS, g = blockIdx.y;

    f32x4 acc[2][NT][NT];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[j][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int b = 0; b < a.B; ++b) {
        const float* __restrict__ xb = a.xz + (int64_t)b * a.T * a.H + g * CG;
        const float* __restrict__ db = a.dc + (int64_t)b * a.T * a.H + g * CG;
        for (int t0 = 0; t0 < a.T; t0 += DW_TT) {
            __syncthreads();
            // x rows t0 + k0 - P ... (XROWS of them), zero outside [0, T); dc rows t0 .. t0 + TT, zero past T
            for (int i = tid; i < XROWS * (CG / 4); i += 256) {
                const int r = i / (CG / 4), c4 = (i % (CG / 4)) * 4;
                const int t = t0 + k0 - a.pad + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t >= 0 && t < a.T) v = *reinterpret_cast<const float4*>(xb + (int64_t)t * a.H + c4);
                float* d = Xs + r * XS + c4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
            for (int i = tid; i < DW_TT * (CG / 4); i += 256) {
                const int r = i / (CG / 4), c4 = (i % (CG / 4)) * 4;
                const int t = t0 + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t < a.T) v = *reinterpret_cast<const float4*>(db + (int64_t)t * a.H + c4);
                float* d = Ds + r * XS + c4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
            __syncthreads();
#pragma unroll 4
            for (int st = 0; st < DW_TT / 4; ++st) {
                float bfr[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) bfr[n] = Ds[(st * 4 + lk) * XS + n * 16 + ln];      // B[k = t][j = co]
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int tap = wave * 2 + j;                                              // local tap
#pragma unroll
                    for (int m = 0; m < NT; ++m) {
                        const float afr = Xs[(st * 4 + lk + tap) * XS + m * 16 + ln];          // A[i = ci][k = t]
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[j][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr, bfr[n], acc[j][m][n], 0, 0, 0);
                    }
                }
            }
        }
    }
    // C/D: col = lane & 15 (co), row = 4 (lane >> 4) + reg (ci)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = k0 + wave * 2 + j;
        if (k >= a.K) continue;
        float* out = a.dwg + ((int64_t)g * a.K + k) * CG * CG;
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(m * 16 + lk * 4 + r) * CG + n * 16 + ln] = acc[j][m][n][r];
    }
}

// ---- training: weight-norm backward, one block per tap ----------------------------------------
// W_eff[k] = g[k] v[k] / ||v[k]||:  dg = <dW, v> / n;  dv = (g / n) (dW - <dW, v> v / n^2)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ wv, const float* __restrict__ wgain,
                                                             const float* __restrict__ dwg, float* __restrict__ dwv,
                                                             float* __restrict__ dwgain, int K, int cg, int H, int groups) {
    __shared__ double red[8];
    const int k = blockIdx.x;
    const int n = cg * H, og = H / groups;
    const float* v = wv + (int64_t)k * n;
    double ss = 0.0, dot = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int ci = i / H, c = i % H, g = c / og, co = c % og;
        const double vv = v[i];
        ss += vv * vv;
        dot += vv * (double)dwg[(((int64_t)g * K + k) * cg + ci) * og + co];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ss += __shfl_xor(ss, off, 64);
        dot += __shfl_xor(dot, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = ss;
        red[4 + (threadIdx.x >> 6)] = dot;
    }
    __syncthreads();
    ss = red[0] + red[1] + red[2] + red[3];
    dot = red[4] + red[5] + red[6] + red[7];
    const double nrm = sqrt(ss > 1e-12 ? ss : 1e-12);
    const double gk = wgain[k];
    if (threadIdx.x == 0) dwgain[k] = (float)(dot / nrm);
    for (int i = threadIdx.x; i < n; i += 256) {
        const int ci = i / H, c = i % H, g = c / og, co = c % og;
        const double dw = dwg[(((int64_t)g * K + k) * cg + ci) * og + co];
        dwv[(int64_t)k * n + i] = (float)(gk / nrm * (dw - dot * (double)v[i] / (nrm * nrm)));
    }
}

template <int CG>
int launch_dw(const PosDwArgs& a, hipStream_t s) {
    constexpr int XS = CG + 2, XROWS = DW_TT + DW_TAPS - 1;
    const size_t lds = (size_t)(((XROWS * XS + 3) & ~3) + DW_TT * XS) * sizeof(float);
    static std::atomic<bool> attr_set{false};   // (idempotent call; atomic so concurrent host threads agree on the flag)
    if (!attr_set) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pos_conv_dw_kernel<CG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid((a.K + DW_TAPS - 1) / DW_TAPS, a.groups), block(256);
    W2V2_LAUNCH(pos_conv_dw_kernel<CG>, grid, block, lds, s, a);
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
    W2V2_LAUNCH(weight_norm_regroup_kernel, dim3(K), dim3(256), 0, s, wv, wg, out, K, cg, H, groups);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_pos_conv(Profiler* prof, const float* x, const float* wg, const float* bias,
                    const int32_t* frame_len, float* y, int B, int T, int H, int K, int groups,
                    int act, hipStream_t s) {
    W2V2_REQUIRE(bias, "pos_conv: null bias");
    return launch_pos_conv_ex(prof, x, wg, bias, frame_len, y, nullptr, B, T, H, K, groups, act, K / 2, 1, s);
}

int launch_pos_conv_ex(Profiler* prof, const float* x, const float* wg, const float* bias,
                       const int32_t* frame_len, float* y, float* pre_act, int B, int T, int H, int K,
                       int groups, int act, int pad_left, int add_residual, hipStream_t s) {
    W2V2_REQUIRE(x && wg && y, "pos_conv: null operand");
    W2V2_REQUIRE(pad_left >= 0 && pad_left < K, "pos_conv: bad left pad %d", pad_left);
    W2V2_REQUIRE(B > 0 && T > 0 && K > 0 && groups > 0 && H % groups == 0, "pos_conv: bad sizes");
    const int cg = H / groups;
    W2V2_REQUIRE(H % 4 == 0, "pos_conv: hidden size must be a multiple of 4");
    PosArgs a{x, wg, bias, frame_len, y, pre_act, B, T, H, K, groups, act, pad_left, add_residual};
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

namespace w2v2 {

int launch_pos_conv_flip_regroup(const float* wg, float* wg_t, int K, int cg, int groups, hipStream_t s) {
    W2V2_REQUIRE(wg && wg_t && K > 0 && cg > 0 && groups > 0, "pos_conv_flip: bad argument");
    W2V2_LAUNCH(flip_regroup_kernel, dim3(1024), dim3(256), 0, s, wg, wg_t, K, cg, groups);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int64_t pos_conv_dw_ws_floats(int, int, int, int, int) { return 8; }   // the reduction stays in registers

int launch_pos_conv_dw(Profiler* prof, const float* xz, const float* dc, float* dwg, float* ws, int B, int T,
                       int H, int K, int groups, hipStream_t s) {
    (void)ws;
    W2V2_REQUIRE(xz && dc && dwg && B > 0 && T > 0 && K > 0 && groups > 0 && H % groups == 0, "pos_conv_dw: bad argument");
    const int cg = H / groups;
    PosDwArgs a{xz, dc, dwg, B, T, H, K, groups, K / 2};
    ProfScope ps(prof, FAM_POSCONV, 2.0 * B * (double)T * H * cg * K, 8.0 * B * (double)T * H * (K / DW_TAPS), s);
    switch (cg) {
        case 16: return launch_dw<16>(a, s);
        case 32: return launch_dw<32>(a, s);
        case 48: return launch_dw<48>(a, s);
        case 64: return launch_dw<64>(a, s);
        default:
            set_error("pos_conv_dw: channels per group = %d unsupported (16, 32, 48, 64)", cg);
            return W2V2_EINVAL;
    }
}

int launch_weight_norm_bwd(const float* wv, const float* wgain, const float* dwg, float* dwv, float* dwgain,
                           int K, int cg, int H, int groups, hipStream_t s) {
    W2V2_REQUIRE(wv && wgain && dwg && dwv && dwgain && K > 0 && cg == H / groups, "weight_norm_bwd: bad argument");
    W2V2_LAUNCH(weight_norm_bwd_kernel, dim3(K), dim3(256), 0, s, wv, wgain, dwg, dwv, dwgain, K, cg, H, groups);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace w2v2

// ======================================================================================
// Precision mode 1: the grouped positional conv as ONE batched GEMM on the bf16 matrix pipe.
// Per (sample, group) the conv is  y[t, n] = sum_{j, c} x[t + j - pad, g cg + c] w[g][j][c][n]; with the group's channels
// packed as P[b][g][r][c] (r = frame + pad, zero rows outside [0, T) and at masked frames) the operand of output frame t
// is the CONTIGUOUS run P[b][g][t * cg .. t * cg + K * cg): a GEMM with overlapping rows (lda = cg < K cg), exactly the
// trick gemm_f32.hip plays for the strided convs, M = T, N = cg, K_gemm = K * cg (6144 for base), batch = B * groups.
// The regrouped kernel (groups, K, cg, og) is already each group's (K cg, og) B matrix; its (og, K cg) bf16 shadow is
// built once per weight change.  x and the effective kernel are rounded to bf16, accumulation is fp32.
// ======================================================================================
namespace w2v2 {
namespace {

// x (B, T, H) fp32 -> P (B, G, Tp, cg) bf16 with Tp = T + K - 1, `pad` zero rows in front; frames >= frame_len[b] are zero.
// Optionally also writes xz (B, T, H) fp32 = the masked x (the residual of the forward pass).
__global__ __launch_bounds__(256) void pos_pack_kernel(const float* __restrict__ x, const int32_t* __restrict__ frame_len,
                                                       uint16_t* __restrict__ P, float* __restrict__ xz, int B, int T, int H,
                                                       int groups, int Tp, int pad) {
    const int cg = H / groups;
    const int64_t total = (int64_t)B * groups * Tp * (cg / 4);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % (cg / 4));
        const int r = (int)((i / (cg / 4)) % Tp);
        const int g = (int)((i / ((int64_t)(cg / 4) * Tp)) % groups);
        const int b = (int)(i / ((int64_t)(cg / 4) * Tp * groups));
        const int t = r - pad;
        const int flen = frame_len ? frame_len[b] : T;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool inside = t >= 0 && t < T;
        if (inside && t < flen) v = *reinterpret_cast<const float4*>(x + ((int64_t)b * T + t) * H + g * cg + 4 * c4);
        *reinterpret_cast<uint2*>(P + (((int64_t)b * groups + g) * Tp + r) * cg + 4 * c4) =
            make_uint2(pack_bf16_rne(v.x, v.y), pack_bf16_rne(v.z, v.w));
        if (xz && inside) *reinterpret_cast<float4*>(xz + ((int64_t)b * T + t) * H + g * cg + 4 * c4) = v;
    }
}

// fp32 variant of the pack (the kernel-gradient GEMM reads A transposed from fp32 and rounds it itself)
__global__ __launch_bounds__(256) void pos_pack32_kernel(const float* __restrict__ x, float* __restrict__ P, int B, int T, int H,
                                                         int groups, int Tp, int pad) {
    const int cg = H / groups;
    const int64_t total = (int64_t)B * groups * Tp * (cg / 4);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % (cg / 4));
        const int r = (int)((i / (cg / 4)) % Tp);
        const int g = (int)((i / ((int64_t)(cg / 4) * Tp)) % groups);
        const int b = (int)(i / ((int64_t)(cg / 4) * Tp * groups));
        const int t = r - pad;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < T) v = *reinterpret_cast<const float4*>(x + ((int64_t)b * T + t) * H + g * cg + 4 * c4);
        *reinterpret_cast<float4*>(P + (((int64_t)b * groups + g) * Tp + r) * cg + 4 * c4) = v;
    }
}

// y = res + act(pre)   (training forward: the pre-activation is kept for the backward, so the GEMM cannot fuse this)
__global__ __launch_bounds__(256) void pos_finish_kernel(const float* __restrict__ pre, const float* __restrict__ res,
                                                         float* __restrict__ y, int64_t n4, int act) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 p = reinterpret_cast<const float4*>(pre)[i];
        float4 o = make_float4(act == 1 ? gelu_erf_fast(p.x) : (act == 2 ? gelu_tanh(p.x) : p.x),
                               act == 1 ? gelu_erf_fast(p.y) : (act == 2 ? gelu_tanh(p.y) : p.y),
                               act == 1 ? gelu_erf_fast(p.z) : (act == 2 ? gelu_tanh(p.z) : p.z),
                               act == 1 ? gelu_erf_fast(p.w) : (act == 2 ? gelu_tanh(p.w) : p.w));
        if (res) {
            const float4 r = reinterpret_cast<const float4*>(res)[i];
            o = make_float4(o.x + r.x, o.y + r.y, o.z + r.z, o.w + r.w);
        }
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

}  // namespace

int64_t pos_conv_bf16_pack_elems(int B, int T, int H, int K) { return (int64_t)B * (T + K - 1) * H; }

// w16: (groups, og, K cg) bf16 shadow of the regrouped kernel wg (groups, K, cg, og)
int launch_pos_conv_weight_shadow(const float* wg, uint16_t* w16, int K, int cg, int groups, hipStream_t s) {
    W2V2_REQUIRE(wg && w16 && K > 0 && cg > 0 && groups > 0, "pos_conv_weight_shadow: bad argument");
    return launch_transpose_to_bf16_batched(wg, w16, K * cg, cg, groups, s);       // one launch for all groups
}

// Same contract as launch_pos_conv_ex.  pack16: pos_conv_bf16_pack_elems() bf16 of scratch; xz_ws: (B, T, H) fp32 of scratch,
// needed only when frame_len && add_residual (the residual is the MASKED input).
int launch_pos_conv_bf16(Profiler* prof, const float* x, const uint16_t* w16, const float* bias, const int32_t* frame_len,
                         float* y, float* pre_act, uint16_t* pack16, float* xz_ws, int B, int T, int H, int K, int groups,
                         int act, int pad_left, int add_residual, hipStream_t s) {
    W2V2_REQUIRE(x && w16 && y && pack16, "pos_conv_bf16: null operand");
    W2V2_REQUIRE(B > 0 && T > 0 && K > 0 && groups > 0 && H % groups == 0 && pad_left >= 0 && pad_left < K, "pos_conv_bf16: bad sizes");
    const int cg = H / groups, Tp = T + K - 1;
    W2V2_REQUIRE(cg % 8 == 0 && cg <= 64 && (K * cg) % 64 == 0, "pos_conv_bf16: channels per group %d / taps %d unsupported", cg, K);
    const bool masked_res = add_residual && frame_len;
    W2V2_REQUIRE(!masked_res || xz_ws, "pos_conv_bf16: the masked residual needs the xz workspace");
    ProfScope ps(prof, FAM_POSCONV, 2.0 * B * (double)T * H * cg * K, 8.0 * B * (double)T * H + 2.0 * K * cg * H, s);
    {
        const int64_t total = (int64_t)B * groups * Tp * (cg / 4);
        int64_t blocks = (total + 255) / 256;
        blocks = blocks > 8192 ? 8192 : blocks;
        W2V2_LAUNCH(pos_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, frame_len, pack16, masked_res ? xz_ws : nullptr, B, T,
                           H, groups, Tp, pad_left);
    }
    const float* res = add_residual ? (masked_res ? xz_ws : x) : nullptr;
    GemmShadows gx;
    gx.A16 = pack16;
    gx.B16 = w16;
    gx.ldb16 = (int64_t)K * cg;
    gx.zmod = groups;
    gx.strideB16 = (int64_t)cg * K * cg;
    gx.strideC2 = (int64_t)T * H;
    gx.strideBias = cg;
    // A: row t of (b, g) starts at pack16 + ((b G + g) Tp + t) cg  -> lda = cg, batch stride Tp cg
    if (pre_act) {
        if (int e = launch_gemm_bf16_x(nullptr, nullptr, cg, (int64_t)Tp * cg, nullptr, cg, 0, pre_act, H, cg, bias, nullptr, T, cg, K * cg,
                                       B * groups, 0, gx, s))
            return e;
        const int64_t n4 = (int64_t)B * T * H / 4;
        int64_t blocks = (n4 + 255) / 256;
        blocks = blocks > 8192 ? 8192 : blocks;
        W2V2_LAUNCH(pos_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pre_act, res, y, n4, act);
        W2V2_HIP_CHECK(hipGetLastError());
        return W2V2_OK;
    }
    return launch_gemm_bf16_x(nullptr, nullptr, cg, (int64_t)Tp * cg, nullptr, cg, 0, y, H, cg, bias, res, T, cg, K * cg, B * groups, act,
                              gx, s);
}

// dwg[g][f = j cg + c][n] = sum_{b, t} xz[b][t + j - pad][g cg + c] dc[b][t][g og + n]: per (sample, group) a GEMM with the
// packed input as a TRANSPOSED, overlapping-row A (element (f, t) at P[t cg + f]) and dc as B; one slab per sample, summed after.
// T % 64 != 0: the contraction runs over Tk = 64 ceil(T / 64) frames, the extra ones contributing exact zeros: dc is copied into
// `dc_pad` (B, Tk, H) with zero tail rows (required then), and the pack holds Tk + K - 1 rows (zeros beyond the padded input).
// pack32: B (Tk + K - 1) H floats.
int launch_pos_conv_dw_bf16(Profiler* prof, const float* xz, const float* dc, float* dwg, float* pack32, float* slabs, float* red_ws,
                            int B, int T, int H, int K, int groups, hipStream_t s, float* dc_pad) {
    W2V2_REQUIRE(xz && dc && dwg && pack32 && slabs, "pos_conv_dw_bf16: null operand");
    const int Tk = (T + 63) / 64 * 64;
    const int cg = H / groups, Tp = Tk + K - 1;
    W2V2_REQUIRE(cg % 4 == 0 && cg <= 64 && B <= 64 && (Tk == T || dc_pad), "pos_conv_dw_bf16: unsupported shape");
    if (Tk != T) {
        W2V2_HIP_CHECK(hipMemcpy2DAsync(dc_pad, (size_t)Tk * H * 4, dc, (size_t)T * H * 4, (size_t)T * H * 4, (size_t)B, hipMemcpyDeviceToDevice, s));
        W2V2_HIP_CHECK(hipMemset2DAsync(dc_pad + (int64_t)T * H, (size_t)Tk * H * 4, 0, (size_t)(Tk - T) * H * 4, (size_t)B, s));
        dc = dc_pad;
    }
    ProfScope ps(prof, FAM_POSCONV, 2.0 * B * (double)T * H * cg * K, 8.0 * B * (double)T * H + 4.0 * (B + 1.0) * K * cg * H, s);
    {
        const int64_t total = (int64_t)B * groups * Tp * (cg / 4);
        int64_t blocks = (total + 255) / 256;
        blocks = blocks > 8192 ? 8192 : blocks;
        W2V2_LAUNCH(pos_pack32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, xz, pack32, B, T, H, groups, Tp, K / 2);
    }
    GemmShadows gx;
    gx.transA = true;
    gx.overlapA = true;
    gx.zmod = groups;
    gx.strideB2 = (int64_t)Tk * H;
    gx.strideC2 = (int64_t)groups * K * cg * cg;
    const int M = K * cg;
    // Round 5: the contraction runs over the frames of SEVERAL samples in one accumulator (GemmShadows::kseg: K = segments of Tk rows, one per
    // sample) instead of one (K cg, og) slab per sample -- 604 MB of slabs written and folded per step at B = 32, and a K of 768 rows per
    // block, most of its time prologue and epilogue.  Slab h of S takes the samples b = h, h + S, h + 2S, ... (so that batch z = h G + g still
    // finds its first operands at z strideA / zo strideB2 + zi strideB), S in {1, 2, 4} chosen to fill whole rounds of the 512 block slots:
    // base 48 tiles x 16 groups x S = 2 -> 1536 blocks, large 64 x 16 x 1 -> 1024.  The sum over a slab's samples is taken in the MFMA
    // accumulators in sample order (deterministic); S = 1 writes the gradient itself.
    if (tune_int("W2V2_POS_DW_KCAT", 1) != 0) {
        const int64_t tiles = (int64_t)((M + 127) / 128) * groups;
        int S = 1;
        double best = 1e30;
        for (int cand : {1, 2, 4}) {
            if (B % cand) continue;
            const int64_t blocks = tiles * cand;
            const double waste = (double)((blocks + 511) / 512 * 512) / (double)blocks;
            if (waste < best - 1e-9) { best = waste; S = cand; }
        }
        gx.kseg = Tk;
        gx.segA = (int64_t)S * groups * Tp * cg;
        gx.segB = (int64_t)S * Tk * H;
        float* dst = S == 1 ? dwg : slabs;
        if (int e = launch_gemm_bf16_x(nullptr, pack32, cg, (int64_t)Tp * cg, dc, H, cg, dst, cg, (int64_t)K * cg * cg, nullptr, nullptr, M, cg,
                                       (B / S) * Tk, S * groups, 0, gx, s))
            return e;
        return S == 1 ? W2V2_OK : launch_colsum(slabs, dwg, S, (int)((int64_t)groups * K * cg * cg), red_ws, 0, s);
    }
    // A^T: element (m = f, k = t) of batch z = b G + g at pack32[z Tp cg + t cg + f]; B (k = t, n) at dc[b Tk H + t H + g og + n]
    if (int e = launch_gemm_bf16_x(nullptr, pack32, cg, (int64_t)Tp * cg, dc, H, cg, slabs, cg, (int64_t)K * cg * cg, nullptr, nullptr, M, cg, Tk,
                                   B * groups, 0, gx, s))
        return e;
    return launch_colsum(slabs, dwg, B, (int)((int64_t)groups * K * cg * cg), red_ws, 0, s);
}

}  // namespace w2v2
