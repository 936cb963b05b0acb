// Layer 0 of the feature extractor: Conv1D(C_in = 1, K = 10, stride 5, valid) ->
// GroupNormalization(groups = C) -> exact GELU, fused so the un-normalised conv
// output (100.8 MB per 246000-sample utterance) is never written.
//
// Reference: feature_extractor.py:31-47,54-59 and tensorflow_addons.py:207-231.
// With groups == channels the "group" norm is a per-(sample, channel) mean /
// population variance over TIME (tf.nn.moments over axis 1), applied as
// tf.nn.batch_normalization does:  y * inv + (beta - mean * inv),  inv = rsqrt(var+eps)*gamma.
//
// The stage is HBM-bound by its single output write.  Two passes over the raw
// waveform (0.98 MB / utterance, L2-resident):
//   pass 1 (stats):  recompute the 10-tap conv in registers, accumulate sum and
//                    sum-of-squares per channel in fp64, one partial per time chunk;
//   finalize:        fp64 combine of the chunk partials -> (scale, shift) per (b, c);
//   pass 2 (apply):  recompute, scale/shift, GELU, write once, coalesced along C.
// A thread owns channels (the kernel taps live in its registers); the waveform
// chunk is staged in LDS and read as a wave-uniform broadcast.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace w2v2 {
namespace {

constexpr int TC = 64;          // frames per block
constexpr int CPT_MAX = 2;      // channels per thread (C <= 512 with 256 threads), looped beyond

struct Conv0Args {
    const float* wave;
    const float* kernel;   // (K, 1, C)
    const float* bias;     // (C) or null
    const float* gamma;
    const float* beta;
    float* out;            // (B, T0, C)
    uint16_t* out16;       // optional bf16 shadow of out (precision mode 1: layer 1's GEMM reads it)
    PlaneOut planes;       // optional planes of out (precision modes bf16x3 / f16x2; conv0_apply4_kernel only)
    double* partial;       // (B, nchunks, 2, C)
    float* scale_shift;    // (B, 2, C)
    const double* ln_const;   // MODE 3 (LayerNorm over channels): LN_CONST doubles, see conv0_ln_const_kernel
    int64_t L;
    int T0, K, stride, C, nchunks, norm_mode, act;
    float eps;
};

// MODE 0: stats, MODE 1: apply (group norm), MODE 2: plain conv(+bias) write
template <int MODE, int KT, int ST>
__global__ __launch_bounds__(256) void conv0_kernel(Conv0Args a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int K = KT > 0 ? KT : a.K;
    const int S = ST > 0 ? ST : a.stride;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int t0 = chunk * TC;
    const int nt = min(TC, a.T0 - t0);
    const int nx = (nt - 1) * S + K;
    const float* __restrict__ wv = a.wave + (int64_t)b * a.L + (int64_t)t0 * S;
    for (int i = threadIdx.x; i < nx; i += 256) xs[i] = wv[i];
    __syncthreads();

    for (int c0 = threadIdx.x; c0 < a.C; c0 += 256 * CPT_MAX) {
        // taps of up to CPT_MAX channels in registers
        float w[CPT_MAX][KT > 0 ? KT : 32];
        float bs[CPT_MAX], sc[CPT_MAX], sh[CPT_MAX];
        double s1[CPT_MAX], s2[CPT_MAX];
#pragma unroll
        for (int j = 0; j < CPT_MAX; ++j) {
            const int c = c0 + 256 * j;
            const bool ok = c < a.C;
#pragma unroll
            for (int k = 0; k < (KT > 0 ? KT : 32); ++k)
                w[j][k] = (ok && k < K) ? a.kernel[(int64_t)k * a.C + c] : 0.f;
            bs[j] = (ok && a.bias) ? a.bias[c] : 0.f;
            s1[j] = 0.0; s2[j] = 0.0;
            if (MODE == 1 && ok) {
                sc[j] = a.scale_shift[((int64_t)b * 2 + 0) * a.C + c];
                sh[j] = a.scale_shift[((int64_t)b * 2 + 1) * a.C + c];
            } else {
                sc[j] = 1.f; sh[j] = 0.f;
            }
        }
        for (int t = 0; t < nt; ++t) {
            const float* xp = xs + t * S;
            float y[CPT_MAX];
#pragma unroll
            for (int j = 0; j < CPT_MAX; ++j) y[j] = bs[j];
            if (KT > 0) {
#pragma unroll
                for (int k = 0; k < (KT > 0 ? KT : 1); ++k) {
                    const float xv = xp[k];
#pragma unroll
                    for (int j = 0; j < CPT_MAX; ++j) y[j] = fmaf(xv, w[j][k], y[j]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    if (k < K) {
                        const float xv = xp[k];
#pragma unroll
                        for (int j = 0; j < CPT_MAX; ++j) y[j] = fmaf(xv, w[j][k], y[j]);
                    }
                }
            }
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < CPT_MAX; ++j) {
                    s1[j] += (double)y[j];
                    s2[j] += (double)y[j] * (double)y[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < CPT_MAX; ++j) {
                    const int c = c0 + 256 * j;
                    if (c < a.C) {
                        float v = MODE == 1 ? apply_act(fmaf(y[j], sc[j], sh[j]), a.act) : y[j];
                        if (a.out) __builtin_nontemporal_store(v, &a.out[((int64_t)b * a.T0 + t0 + t) * a.C + c]);   // 3.2 GB streamed once
                        if (a.out16) a.out16[((int64_t)b * a.T0 + t0 + t) * a.C + c] = (uint16_t)pack_bf16_rne(v, 0.f);
                    }
                }
            }
        }
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < CPT_MAX; ++j) {
                const int c = c0 + 256 * j;
                if (c < a.C) {
                    double* p = a.partial + (((int64_t)b * a.nchunks + chunk) * 2) * a.C;
                    p[c] = s1[j];
                    p[a.C + c] = s2[j];
                }
            }
        }
    }
}

// Apply / plain-conv pass with 16-byte stores: a lane owns FOUR consecutive channels (40 taps in registers for K = 10), so
// a wave-level store is 64 x 16 B = 1 KiB of one output row instead of 256 B, and the bf16 shadow goes out as 8-byte
// stores.  C / 4 lanes cover a frame; the block's 256 threads work on 256 / (C / 4) frames at a time (2 for C = 512).
// The stage is bound by its single 3.2 GB output write (B = 32 x 246000): what matters is how few, how wide and how
// regular the store instructions are.  MODE 1: scale/shift + activation (group norm), MODE 2: plain conv(+bias).
using f32x4_c0 = __attribute__((ext_vector_type(4))) float;
using u32x2_c0 = __attribute__((ext_vector_type(2))) unsigned;

// MODE 3: LayerNorm over the C channels of each frame + activation (the robust / xlsr extractor, feature_extractor.py:40-47 with
// layer norm).  A frame's channels are y_c = w_c . x + b_c over the SAME 10 samples x, so its moments over c need no pass over y:
//   mean_c y = wbar . x + bbar,     var_c y = x^T Cw x + 2 cwb . x + varb
// with wbar = mean_c w_c, Cw = mean_c (w_c - wbar)(w_c - wbar)^T (10 x 10), cwb = mean_c (w_c - wbar)(b_c - bbar), varb = var_c b:
// 122 numbers of the layer's kernel (conv0_ln_const_kernel, fp64).  One thread per frame evaluates them in fp64 (65 FMAs, against
// 5120 for the frame's taps) before the apply loop, and the un-normalised conv output -- 3.1 GB at 16 x 480000 -- is never
// written or re-read.  Centred weights: the quadratic form has no cancellation to lose.
constexpr int LN_CONST = 10 + 1 + 100 + 10 + 1;     // wbar | bbar | Cw | cwb | varb   (K = 10)

__global__ __launch_bounds__(128) void conv0_ln_const_kernel(const float* __restrict__ kernel /* (10, C) */, const float* __restrict__ bias,
                                                             double* __restrict__ out, int C) {
    __shared__ double wbar[10], bbar;
    const int t = threadIdx.x;
    if (t < 10) {
        double sum = 0.0;
        for (int c = 0; c < C; ++c) sum += (double)kernel[t * C + c];
        wbar[t] = sum / C;
    } else if (t == 10) {
        double sum = 0.0;
        for (int c = 0; c < C; ++c) sum += bias ? (double)bias[c] : 0.0;
        bbar = sum / C;
    }
    __syncthreads();
    if (t < 10) out[t] = wbar[t];
    if (t == 10) out[10] = bbar;
    if (t < 100) {
        const int i = t / 10, j = t % 10;
        double sum = 0.0;
        for (int c = 0; c < C; ++c) sum += ((double)kernel[i * C + c] - wbar[i]) * ((double)kernel[j * C + c] - wbar[j]);
        out[11 + t] = sum / C;
    } else if (t < 110) {
        const int i = t - 100;
        double sum = 0.0;
        for (int c = 0; c < C; ++c) sum += ((double)kernel[i * C + c] - wbar[i]) * ((bias ? (double)bias[c] : 0.0) - bbar);
        out[111 + i] = sum / C;
    } else if (t == 110) {
        double sum = 0.0;
        for (int c = 0; c < C; ++c) { const double d = (bias ? (double)bias[c] : 0.0) - bbar; sum += d * d; }
        out[121] = sum / C;
    }
}

template <int MODE, int KT, int ST>
__global__ __launch_bounds__(256) void conv0_apply4_kernel(Conv0Args a, int tpr /* lanes per frame = C / 4 */) {
    constexpr int TCB = TC;      // frames per block: 32 / 128 / 256 / 512 measured 0.644 / 0.638 / 0.655 / 0.711 ms against 0.627-0.635 for 64
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int t0 = chunk * TCB;
    const int nt = min(TCB, a.T0 - t0);
    const int nx = (nt - 1) * ST + KT;
    const float* __restrict__ wv = a.wave + (int64_t)b * a.L + (int64_t)t0 * ST;
    for (int i = threadIdx.x; i < nx; i += 256) xs[i] = wv[i];
    const int q = threadIdx.x % tpr, fp = threadIdx.x / tpr, fpb = 256 / tpr, c = 4 * q;
    f32x4_c0 w[KT], bs = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KT; ++k) w[k] = *reinterpret_cast<const f32x4_c0*>(a.kernel + (int64_t)k * a.C + c);
    if (a.bias) bs = *reinterpret_cast<const f32x4_c0*>(a.bias + c);
    if (MODE == 1) {
        sc = *reinterpret_cast<const f32x4_c0*>(a.scale_shift + ((int64_t)b * 2 + 0) * a.C + c);
        sh = *reinterpret_cast<const f32x4_c0*>(a.scale_shift + ((int64_t)b * 2 + 1) * a.C + c);
    }
    if (MODE == 3) {      // gamma / beta of this lane's channels
        sc = *reinterpret_cast<const f32x4_c0*>(a.gamma + c);
        sh = *reinterpret_cast<const f32x4_c0*>(a.beta + c);
    }
    __syncthreads();
    __shared__ float fr_mean[TC], fr_rstd[TC];
    if (MODE == 3) {
        static_assert(KT == 10 || MODE != 3, "the LayerNorm constants are laid out for 10 taps");
        if ((int)threadIdx.x < nt) {
            const float* xp = xs + threadIdx.x * ST;
            const double* __restrict__ k = a.ln_const;
            double x[KT > 0 ? KT : 1], mean = k[10], var = k[121];
#pragma unroll
            for (int i = 0; i < KT; ++i) x[i] = (double)xp[i];
#pragma unroll
            for (int i = 0; i < KT; ++i) {
                mean = fma(k[i], x[i], mean);
                double r = 2.0 * k[111 + i];
#pragma unroll
                for (int j = 0; j < KT; ++j) r = fma(k[11 + i * 10 + j], x[j], r);
                var = fma(r, x[i], var);
            }
            fr_mean[threadIdx.x] = (float)mean;
            fr_rstd[threadIdx.x] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)a.eps));
        }
        __syncthreads();
    }
    float* __restrict__ orow = a.out ? a.out + ((int64_t)b * a.T0 + t0) * a.C + c : nullptr;
    uint16_t* __restrict__ orow16 = a.out16 ? a.out16 + ((int64_t)b * a.T0 + t0) * a.C + c : nullptr;
    uint16_t* __restrict__ prow = a.planes.p ? a.planes.p + ((int64_t)b * a.T0 + t0) * a.C + c : nullptr;
    bool ovf = false;
    for (int t = fp; t < nt; t += fpb) {
        const float* xp = xs + t * ST;
        f32x4_c0 y = bs;
        if (MODE == 1 && a.act == 3) {   // (MODE 3 takes the scalar path: its normalisation is per frame, not per channel)
            // bf16 mode: this stage writes half the bytes and becomes VALU-bound, so the taps, the normalisation and the GELU
            // (gelu_erf_fast, common.h) run two channels per instruction (v_pk_fma_f32 / v_pk_mul_f32)
            using f2 = __attribute__((ext_vector_type(2))) float;
            f2 ya = {bs[0], bs[1]}, yb = {bs[2], bs[3]};
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const float xv = xp[k];
                const f2 xx = {xv, xv};
                ya = __builtin_elementwise_fma(xx, f2{w[k][0], w[k][1]}, ya);
                yb = __builtin_elementwise_fma(xx, f2{w[k][2], w[k][3]}, yb);
            }
            ya = gelu_erf_fast2(__builtin_elementwise_fma(ya, f2{sc[0], sc[1]}, f2{sh[0], sh[1]}));
            yb = gelu_erf_fast2(__builtin_elementwise_fma(yb, f2{sc[2], sc[3]}, f2{sh[2], sh[3]}));
            y = f32x4_c0{ya[0], ya[1], yb[0], yb[1]};
        } else {
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const float xv = xp[k];
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = fmaf(xv, w[k][j], y[j]);
            }
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = apply_act(fmaf(y[j], sc[j], sh[j]), a.act);
            }
            if (MODE == 3) {      // (same association as layer_norm_kernel: ((y - mean) rstd) gamma + beta)
                const float mean = fr_mean[t], rstd = fr_rstd[t];
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = apply_act((y[j] - mean) * rstd * sc[j] + sh[j], a.act);
            }
        }
        // nontemporal: 0.635 ms against 0.657 with plain stores (3.2 GB streamed once; tools/write_bw.hip: a bare 128-KiB-per-block
        // fill reaches 5.7-5.9 TB/s, hipMemsetAsync 6.5, this kernel 5.1-5.2 including the waveform reads and the arithmetic)
        if (orow) __builtin_nontemporal_store(y, reinterpret_cast<f32x4_c0*>(orow + (int64_t)t * a.C));
        if (orow16) {
            u32x2_c0 h;
            h[0] = pack_bf16_rne(y[0], y[1]);
            h[1] = pack_bf16_rne(y[2], y[3]);
            *reinterpret_cast<u32x2_c0*>(orow16 + (int64_t)t * a.C) = h;
        }
        if (prow) store_planes4(prow + (int64_t)t * a.C, a.planes.plane, a.planes.fmt, y, ovf);
    }
    report_overflow(a.planes.range_flag, ovf);
}

// (b, c): fp64 combine of chunk partials -> scale = rsqrt(var+eps)*gamma, shift = beta - mean*scale
__global__ void conv0_finalize_kernel(Conv0Args a, int B) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * a.C) return;
    const int b = (int)(i / a.C), c = (int)(i % a.C);
    double s1 = 0.0, s2 = 0.0;
    for (int ch = 0; ch < a.nchunks; ++ch) {
        const double* p = a.partial + (((int64_t)b * a.nchunks + ch) * 2) * a.C;
        s1 += p[c];
        s2 += p[a.C + c];
    }
    const double n = (double)a.T0;
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double inv = (double)a.gamma[c] / sqrt(var + (double)a.eps);
    a.scale_shift[((int64_t)b * 2 + 0) * a.C + c] = (float)inv;
    a.scale_shift[((int64_t)b * 2 + 1) * a.C + c] = (float)((double)a.beta[c] - mean * inv);
}

// ---- GroupNorm statistics without computing the conv (K = 10) ---------------------------------------------------
// sum_t y[t,c] and sum_t y[t,c]^2 of  y[t,c] = bias_c + sum_k w[k,c] x[S t + k]  are a linear and a quadratic form in
//   X1[k] = sum_t x[S t + k]              (K numbers per sample)
//   R[k,k'] = sum_t x[S t + k] x[S t + k']  (K (K+1) / 2 numbers per sample: the strided autocorrelation of the waveform)
// so the statistics of all 512 channels cost 65 accumulators per frame instead of 512 x 10 multiply-adds: the pass
// that recomputed the whole conv (0.27 ms) and the 201 MB of per-chunk fp64 partials it fed to the finalize kernel
// (0.30 ms) become 2.7 M multiply-adds per sample and 65 doubles per block.  Products are exact in fp64 of fp32 inputs;
// a thread sums 8 of them in fp32 before everything else is carried in fp64.
constexpr int GK = 10, GN = GK + GK * (GK + 1) / 2;   // 65
constexpr int GFR = 2048;                              // frames per block

__global__ __launch_bounds__(256) void conv0_gram_kernel(Conv0Args a, int nblk) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    __shared__ double red[4][GN];
    const int b = blockIdx.y, blk = blockIdx.x, S = a.stride;
    const int t0 = blk * GFR, nt = min(GFR, a.T0 - t0), nx = (nt - 1) * S + GK;
    const float* __restrict__ wv = a.wave + (int64_t)b * a.L + (int64_t)t0 * S;
    for (int i = threadIdx.x; i < nx; i += 256) xs[i] = wv[i];
    __syncthreads();
    float acc[GN];
#pragma unroll
    for (int i = 0; i < GN; ++i) acc[i] = 0.f;
    for (int t = threadIdx.x; t < nt; t += 256) {
        float xv[GK];
#pragma unroll
        for (int k = 0; k < GK; ++k) xv[k] = xs[t * S + k];
        int n = GK;
#pragma unroll
        for (int k = 0; k < GK; ++k) {
            acc[k] += xv[k];
#pragma unroll
            for (int k2 = k; k2 < GK; ++k2) {
                acc[n] = fmaf(xv[k], xv[k2], acc[n]);
                ++n;
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < GN; ++i) {
        double v = (double)acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < GN)
        a.partial[((int64_t)b * nblk + blk) * GN + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// one block per sample: combine the Gram partials, then every channel evaluates its two forms in fp64
__global__ __launch_bounds__(256) void conv0_gram_finalize_kernel(Conv0Args a, int nblk) {
    __shared__ double g[GN];
    const int b = blockIdx.x;
    if (threadIdx.x < GN) {
        double v = 0.0;
        for (int i = 0; i < nblk; ++i) v += a.partial[((int64_t)b * nblk + i) * GN + threadIdx.x];
        g[threadIdx.x] = v;
    }
    __syncthreads();
    const double n = (double)a.T0;
    for (int c = threadIdx.x; c < a.C; c += 256) {
        double w[GK];
#pragma unroll
        for (int k = 0; k < GK; ++k) w[k] = (double)a.kernel[(int64_t)k * a.C + c];
        double lin = 0.0, quad = 0.0;
        int idx = GK;
#pragma unroll
        for (int k = 0; k < GK; ++k) {
            lin += w[k] * g[k];
#pragma unroll
            for (int k2 = k; k2 < GK; ++k2) quad += (k2 == k ? 1.0 : 2.0) * w[k] * w[k2] * g[idx++];
        }
        const double bias = a.bias ? (double)a.bias[c] : 0.0;
        const double s1 = lin + n * bias, s2 = quad + 2.0 * bias * lin + n * bias * bias;
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const double inv = (double)a.gamma[c] / sqrt(var + (double)a.eps);
        a.scale_shift[((int64_t)b * 2 + 0) * a.C + c] = (float)inv;
        a.scale_shift[((int64_t)b * 2 + 1) * a.C + c] = (float)((double)a.beta[c] - mean * inv);
    }
}

template <int MODE>
void launch_mode(const Conv0Args& a, int B, hipStream_t s) {
    dim3 grid(a.nchunks, B), block(256);
    const size_t lds = ((size_t)(TC - 1) * a.stride + a.K + 4) * sizeof(float);
    // the 16-byte-store kernel: C / 4 lanes per frame must tile the block, and every pointer it vectorises must be aligned
    const int tpr = a.C / 4;
    const bool vec_ok = MODE != 0 && a.K == 10 && a.stride == 5 && a.C % 4 == 0 && tpr >= 1 && tpr <= 256 && 256 % tpr == 0 &&
                        ((reinterpret_cast<uintptr_t>(a.out) | reinterpret_cast<uintptr_t>(a.kernel) |
                          reinterpret_cast<uintptr_t>(a.bias) | (MODE == 1 ? reinterpret_cast<uintptr_t>(a.scale_shift) : 0)) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(a.out16) & 7) == 0;
    if constexpr (MODE != 0) {
        if (vec_ok) {
            W2V2_LAUNCH((conv0_apply4_kernel<MODE, 10, 5>), grid, block, lds, s, a, tpr);
            return;
        }
    }
    if (a.K == 10 && a.stride == 5)
        W2V2_LAUNCH((conv0_kernel<MODE, 10, 5>), grid, block, lds, s, a);
    else
        W2V2_LAUNCH((conv0_kernel<MODE, 0, 0>), grid, block, lds, s, a);
}

}  // namespace

static inline int conv0_nchunks(int64_t L, int K, int stride) {
    const int64_t T0 = 1 + (L - K) / stride;
    return (int)((T0 + TC - 1) / TC);
}

int64_t conv0_ws_floats(int B, int64_t L, int K, int stride, int C) {
    if (L < K || stride <= 0) return 0;
    const int64_t nch = conv0_nchunks(L, K, stride);
    return 2 * ((int64_t)B * nch * 2 * C) /* fp64 partials */ + (int64_t)B * 2 * C + 8 + 2 * LN_CONST /* LayerNorm-mode constants */;
}

int launch_conv0(Profiler* prof, const float* wave, const float* kernel, const float* bias,
                 const float* gamma, const float* beta, float* out, float* ws, int B, int64_t L,
                 int K, int stride, int C, float eps, int norm_mode, int act, hipStream_t s) {
    return launch_conv0_x(prof, wave, kernel, bias, gamma, beta, out, nullptr, ws, B, L, K, stride, C, eps, norm_mode, act, s);
}

int launch_conv0_x(Profiler* prof, const float* wave, const float* kernel, const float* bias,
                   const float* gamma, const float* beta, float* out, uint16_t* out16, float* ws, int B, int64_t L,
                   int K, int stride, int C, float eps, int norm_mode, int act, hipStream_t s, const PlaneOut* planes) {
    const PlaneOut pl = planes ? *planes : PlaneOut{};
    W2V2_REQUIRE(wave && kernel && (out || out16 || pl.p), "conv0: null operand");
    // planes come out of the 16-byte-store kernel only (K = 10, stride 5, C / 4 lanes tiling the block, aligned operands); other
    // geometries: write fp32 and split it (launch_split_planes)
    W2V2_REQUIRE(!pl.p || (K == 10 && stride == 5 && C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0 && pl.plane % 4 == 0 &&
                           (reinterpret_cast<uintptr_t>(pl.p) & 7) == 0 &&
                           ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(kernel) | reinterpret_cast<uintptr_t>(bias) |
                             reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0),
                 "conv0: plane output needs the K = 10 / stride 5 geometry, C %% 4 == 0 and 16-byte aligned operands");
    W2V2_REQUIRE(B > 0 && C > 0 && K > 0 && K <= 32 && stride > 0 && L >= K,
                 "conv0: unsupported B=%d C=%d K=%d stride=%d L=%lld", B, C, K, stride, (long long)L);
    W2V2_REQUIRE(norm_mode >= 0 && norm_mode <= 2, "conv0: bad norm_mode %d", norm_mode);
    Conv0Args a;
    a.wave = wave; a.kernel = kernel; a.bias = bias; a.gamma = gamma; a.beta = beta; a.out = out; a.out16 = out16; a.planes = pl;
    a.L = L; a.K = K; a.stride = stride; a.C = C; a.eps = eps; a.norm_mode = norm_mode; a.act = act;
    a.T0 = (int)(1 + (L - K) / stride);
    a.nchunks = conv0_nchunks(L, K, stride);
    const double out_bytes = ((out ? 4.0 : 0.0) + (out16 ? 2.0 : 0.0) + (pl.p ? 2.0 * plane_count(pl.fmt) : 0.0)) * B * (double)a.T0 * C, in_bytes = 4.0 * B * (double)L;
    const double flops = 2.0 * B * (double)a.T0 * C * K;
    if (norm_mode == 1) {
        W2V2_REQUIRE(!pl.p || (reinterpret_cast<uintptr_t>(out16) & 7) == 0, "conv0: plane output needs an 8-byte aligned bf16 output");
        ProfScope ps(prof, FAM_CONV0_APPLY, flops, in_bytes + out_bytes, s);
        launch_mode<2>(a, B, s);
        W2V2_HIP_CHECK(hipGetLastError());
        return W2V2_OK;
    }
    if (norm_mode == 2) {       // conv -> LayerNorm over channels -> activation, one pass (conv0_apply4_kernel MODE 3)
        W2V2_REQUIRE(ws && gamma && beta, "conv0: layer-norm mode needs workspace, gamma and beta");
        const int tpr = C / 4;
        const bool vec_ok = K == 10 && stride == 5 && C % 4 == 0 && tpr >= 1 && tpr <= 256 && 256 % tpr == 0 &&
                            ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(kernel) | reinterpret_cast<uintptr_t>(bias) |
                              reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0 &&
                            (reinterpret_cast<uintptr_t>(out16) & 7) == 0;
        if (!vec_ok) {          // other geometries: the plain conv, then the LayerNorm kernel in place
            W2V2_REQUIRE(out && !pl.p, "conv0: this geometry needs the fp32 output buffer for its LayerNorm pass (and cannot write planes)");
            a.out16 = nullptr;
            {
                ProfScope ps(prof, FAM_CONV0_APPLY, flops, in_bytes + 4.0 * B * (double)a.T0 * C, s);
                launch_mode<2>(a, B, s);
            }
            W2V2_HIP_CHECK(hipGetLastError());
            return launch_layer_norm_x(prof, out, out, gamma, beta, (int64_t)B * a.T0, C, eps, act, out16, s);
        }
        double* kc = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(ws) + 7) & ~(uintptr_t)7);
        a.ln_const = kc;
        ProfScope ps(prof, FAM_CONV0_APPLY, flops, in_bytes + out_bytes, s);
        W2V2_LAUNCH(conv0_ln_const_kernel, dim3(1), dim3(128), 0, s, kernel, bias, kc, C);
        const size_t lds = ((size_t)(TC - 1) * stride + K + 4) * sizeof(float);
        W2V2_LAUNCH((conv0_apply4_kernel<3, 10, 5>), dim3(a.nchunks, B), dim3(256), lds, s, a, tpr);
        W2V2_HIP_CHECK(hipGetLastError());
        return W2V2_OK;
    }
    W2V2_REQUIRE(ws && gamma && beta, "conv0: group-norm mode needs workspace, gamma and beta");
    // workspace: 8-byte aligned fp64 partials first, then fp32 scale/shift
    uintptr_t p = (reinterpret_cast<uintptr_t>(ws) + 7) & ~(uintptr_t)7;
    a.partial = reinterpret_cast<double*>(p);
    a.scale_shift = reinterpret_cast<float*>(a.partial + (int64_t)B * a.nchunks * 2 * C);
    {
        ProfScope ps(prof, FAM_CONV0_STATS, flops, in_bytes, s);
        if (K == GK) {
            const int nblk = (a.T0 + GFR - 1) / GFR;
            const size_t lds = ((size_t)(GFR - 1) * stride + GK + 4) * sizeof(float);
            if (lds <= 60 * 1024) {
                W2V2_LAUNCH(conv0_gram_kernel, dim3(nblk, B), dim3(256), lds, s, a, nblk);
                W2V2_LAUNCH(conv0_gram_finalize_kernel, dim3(B), dim3(256), 0, s, a, nblk);
            } else {
                launch_mode<0>(a, B, s);
                const int64_t n = (int64_t)B * C;
                W2V2_LAUNCH(conv0_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, B);
            }
        } else {
            launch_mode<0>(a, B, s);
            const int64_t n = (int64_t)B * C;
            W2V2_LAUNCH(conv0_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, B);
        }
    }
    // (the plane output exists only in the 16-byte-store kernel, and launch_mode<1> also tests the alignment of the
    //  scale / shift table and of the bf16 output before it picks that kernel -- the generic kernel would leave the planes unwritten)
    W2V2_REQUIRE(!pl.p || ((reinterpret_cast<uintptr_t>(a.scale_shift) & 15) == 0 && (reinterpret_cast<uintptr_t>(out16) & 7) == 0),
                 "conv0: plane output needs a 16-byte aligned workspace and an 8-byte aligned bf16 output");
    {
        ProfScope ps(prof, FAM_CONV0_APPLY, flops, in_bytes + out_bytes, s);
        launch_mode<1>(a, B, s);
    }
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace w2v2
