// Row LayerNorm over the channel axis (+ optional GELU), fp32.
//
// tf.keras.layers.LayerNormalization(axis=-1, epsilon=eps) as the reference calls
// it (feature_extractor.py:48-50,86-88; encoder.py:96-98,105-108,232-234):
// population variance, y = (x - mean) * rsqrt(var + eps) * gamma + beta.
//
// HBM-bound: one read + one write of the row.  One wave64 per row; the row is
// held in registers (float4 per lane per 256-channel chunk) so mean and the
// centred variance are two in-register passes -- no E[x^2]-mean^2 cancellation.
#include <stdlib.h>

#include "common.h"
#include "train.h"

namespace w2v2 {
namespace {

// One wave walks rows  first, first + stride, ...  with the NEXT row's loads issued before the current row is reduced,
// so a wave always has a 3-KiB row in flight: a one-row-per-wave launch (24.5 k short-lived waves for a (24576, 768)
// tensor) spent more time starting waves than moving data (2.85 TB/s); this form is bandwidth-bound.
template <int NV>   // NV float4 per lane: covers C <= NV * 256
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* __restrict__ x,
                                                         float* __restrict__ y,
                                                         uint16_t* __restrict__ y16,   // optional bf16 shadow of y
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         int64_t rows, int C, float eps, int act, PlaneOut pl) {
    const int lane = threadIdx.x & 63;
    bool ovf = false;
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bool vec = (C & 3) == 0;
    auto load_row = [&](int64_t r, float4 (&v)[NV]) {
        const float* xr = x + r * C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (vec && c < C) {
                v[i] = *reinterpret_cast<const float4*>(xr + c);
            } else {
                v[i].x = c + 0 < C ? xr[c + 0] : 0.f;
                v[i].y = c + 1 < C ? xr[c + 1] : 0.f;
                v[i].z = c + 2 < C ? xr[c + 2] : 0.f;
                v[i].w = c + 3 < C ? xr[c + 3] : 0.f;
            }
        }
    };
    // gamma / beta of this lane's columns stay in registers across rows
    float4 gv[NV], bv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        gv[i] = make_float4(c + 0 < C ? gamma[c + 0] : 0.f, c + 1 < C ? gamma[c + 1] : 0.f, c + 2 < C ? gamma[c + 2] : 0.f,
                            c + 3 < C ? gamma[c + 3] : 0.f);
        bv[i] = make_float4(c + 0 < C ? beta[c + 0] : 0.f, c + 1 < C ? beta[c + 1] : 0.f, c + 2 < C ? beta[c + 2] : 0.f,
                            c + 3 < C ? beta[c + 3] : 0.f);
    }
    float4 v[NV], nx[NV];
    load_row(row, v);
    while (true) {
        const int64_t next = row + stride;
        const bool more = next < rows;
        load_row(more ? next : row, nx);          // unconditional prefetch (re-reads the last row once): stays in registers
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = wave_sum(sum) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            const float dx = c + 0 < C ? v[i].x - mean : 0.f;
            const float dy = c + 1 < C ? v[i].y - mean : 0.f;
            const float dz = c + 2 < C ? v[i].z - mean : 0.f;
            const float dw = c + 3 < C ? v[i].w - mean : 0.f;
            v[i] = make_float4(dx, dy, dz, dw);
            sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)C + eps);
        float* yr = y ? y + row * C : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c >= C) continue;
            float o[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            const float g4[4] = {gv[i].x, gv[i].y, gv[i].z, gv[i].w}, b4[4] = {bv[i].x, bv[i].y, bv[i].z, bv[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < C) o[e] = apply_act(o[e] * rstd * g4[e] + b4[e], act);
            if (y16) {      // nearest-even bf16 copy for the consumer GEMM (precision mode 1)
                uint16_t* hr = y16 + row * C + c;
                if (vec) {
                    *reinterpret_cast<uint2*>(hr) = make_uint2(pack_bf16_rne(o[0], o[1]), pack_bf16_rne(o[2], o[3]));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < C) hr[e] = (uint16_t)pack_bf16_rne(o[e], 0.f);
                }
            }
            if (pl.p) store_planes4(pl.p + row * C + c, pl.plane, pl.fmt, f32x4_t{o[0], o[1], o[2], o[3]}, ovf);      // (launcher: C % 4 == 0)
            if (!y) continue;       // bf16-only output: the consumer streams the shadow
            if (vec) {
                *reinterpret_cast<float4*>(yr + c) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < C) yr[c + e] = o[e];
            }
        }
        if (!more) break;
        row = next;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = nx[i];
    }
    report_overflow(pl.range_flag, ovf);
}


// t1 = dropout(a) + res;  y = LayerNorm(t1)  in ONE pass (training forward, encoder.py:116-119 / 122-124: "x + drop(attn)" followed by
// the layer's next LayerNorm).  Round 4: as two kernels the row was written by the dropout pass and read again by the LayerNorm
// (414 MB of traffic per layer at B = 32; now 339).  Same element arithmetic in the same order as dropout_fwd_kernel (keep ? a / (1 - p)
// : 0, then + res) and layer_norm_kernel, so the results are bit-identical to the two-kernel form.  C % 4 == 0, 16-byte aligned rows.
using ln_f32x4 = __attribute__((ext_vector_type(4))) float;
template <int NV, bool DROP>      // DROP = false: plain LayerNorm of `a` (res, t1, p unused) -- the same row loop, serving launch_layer_norm_x
__global__ __launch_bounds__(256) void layer_norm_drop_kernel(const float* __restrict__ a, const float* __restrict__ res, float* __restrict__ t1,
                                                              float* __restrict__ y, uint16_t* __restrict__ y16, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int64_t rows, int C, float eps, float p, uint64_t seed,
                                                              uint32_t stream, PlaneOut pl) {
    const int lane = threadIdx.x & 63;
    bool ovf = false;
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float inv = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const uint32_t key = dropout_key(seed, stream), thr = dropout_threshold(p);
    const ln_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    ln_f32x4 gv[NV], bv[NV], va[NV], vr[NV], v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        gv[i] = c < C ? *reinterpret_cast<const ln_f32x4*>(gamma + c) : zero;
        bv[i] = c < C ? *reinterpret_cast<const ln_f32x4*>(beta + c) : zero;
        va[i] = c < C ? *reinterpret_cast<const ln_f32x4*>(a + row * C + c) : zero;
        vr[i] = (DROP && c < C) ? *reinterpret_cast<const ln_f32x4*>(res + row * C + c) : zero;
    }
    while (true) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            ln_f32x4 e = va[i];
            if (DROP && p > 0.f && c < C) {
                const uint32_t pair = (uint32_t)((uint64_t)row * (uint64_t)C + (uint64_t)c) >> 1;      // (index mod 2^32) >> 1, as dropout_fwd_kernel
                const uint32_t w0 = dropout_word(key, pair), w1 = dropout_word(key, pair + 1u);
                e[0] = dropout_keep_lo(w0, thr) ? e[0] * inv : 0.0f;
                e[1] = dropout_keep_hi(w0, thr) ? e[1] * inv : 0.0f;
                e[2] = dropout_keep_lo(w1, thr) ? e[2] * inv : 0.0f;
                e[3] = dropout_keep_hi(w1, thr) ? e[3] * inv : 0.0f;
            }
            v[i] = DROP ? e + vr[i] : e;
            if (DROP && c < C) *reinterpret_cast<ln_f32x4*>(t1 + row * C + c) = v[i];
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
        // the next row's loads go out now, into the registers just consumed: they are in flight under the two reductions and the stores
        const int64_t next = row + stride;
        const bool more = next < rows;
        const int64_t nrow = more ? next : row;          // (unconditional: the last trip re-reads its own row)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            va[i] = c < C ? *reinterpret_cast<const ln_f32x4*>(a + nrow * C + c) : zero;
            if (DROP) vr[i] = c < C ? *reinterpret_cast<const ln_f32x4*>(res + nrow * C + c) : zero;
        }
        const float mean = wave_sum(sum) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            const ln_f32x4 d = c < C ? v[i] - mean : zero;
            v[i] = d;
            sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c >= C) continue;
            ln_f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = v[i][k] * rstd * gv[i][k] + bv[i][k];
            if (y16) *reinterpret_cast<uint2*>(y16 + row * C + c) = make_uint2(pack_bf16_rne(o[0], o[1]), pack_bf16_rne(o[2], o[3]));
            if (pl.p) store_planes4(pl.p + row * C + c, pl.plane, pl.fmt, o, ovf);
            if (y) *reinterpret_cast<ln_f32x4*>(y + row * C + c) = o;
        }
        if (!more) break;
        row = next;
    }
    report_overflow(pl.range_flag, ovf);
}

}  // namespace

// (the index enters the dropout hash modulo 2^32, exactly as in dropout_fwd_kernel; callers keep rows * C below 2^32 or accept the
//  wrap, as there)
int launch_layer_norm_drop(Profiler* prof, const float* a, const float* res, float* t1, float* y, uint16_t* y16, const float* gamma,
                           const float* beta, int64_t rows, int C, float eps, float p, uint64_t seed, uint32_t stream, hipStream_t s) {
    W2V2_REQUIRE(a && res && t1 && (y || y16) && gamma && beta, "layer_norm_drop: null operand");
    W2V2_REQUIRE(rows > 0 && C > 0 && C <= 2048 && (C & 3) == 0 && p >= 0.f && p < 1.f, "layer_norm_drop: rows=%lld C=%d p=%f unsupported", (long long)rows, C, p);
    W2V2_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(t1) | reinterpret_cast<uintptr_t>(y) |
                   reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0 && (reinterpret_cast<uintptr_t>(y16) & 7) == 0,
                 "layer_norm_drop: unaligned operand");
    const int64_t want = (rows + 3) / 4;
    const int cap = tune_int("W2V2_LN_BLOCKS", 256 * 4);
    dim3 grid((unsigned)(want < cap ? want : cap)), block(256);
    ProfScope ps(prof, FAM_LAYERNORM, 8.0 * rows * C, (12.0 + (y ? 4.0 : 0.0) + (y16 ? 2.0 : 0.0)) * rows * C, s);
    // (two instances only: hipcc 7.2 crashes in its machine-copy-propagation pass on the <2> instance of this kernel, and the encoder widths
    //  this pass serves are 768 / 1024; narrower rows -- the tiny test configurations -- leave the upper lanes of the <4> instance idle)
    if (C <= 1024)
        W2V2_LAUNCH((layer_norm_drop_kernel<4, true>), grid, block, 0, s, a, res, t1, y, y16, gamma, beta, rows, C, eps, p, seed, stream, PlaneOut{});
    else
        W2V2_LAUNCH((layer_norm_drop_kernel<8, true>), grid, block, 0, s, a, res, t1, y, y16, gamma, beta, rows, C, eps, p, seed, stream, PlaneOut{});
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_layer_norm(Profiler* prof, const float* x, float* y, const float* gamma,
                      const float* beta, int64_t rows, int C, float eps, int act, hipStream_t s) {
    return launch_layer_norm_x(prof, x, y, gamma, beta, rows, C, eps, act, nullptr, s);
}

int launch_layer_norm_x(Profiler* prof, const float* x, float* y, const float* gamma, const float* beta, int64_t rows,
                        int C, float eps, int act, uint16_t* y16, hipStream_t s, const PlaneOut* planes) {
    const PlaneOut pl = planes ? *planes : PlaneOut{};
    W2V2_REQUIRE(x && (y || y16 || pl.p) && gamma && beta, "layer_norm: null operand");
    W2V2_REQUIRE(!pl.p || ((C & 3) == 0 && pl.plane % 4 == 0 && (reinterpret_cast<uintptr_t>(pl.p) & 7) == 0), "layer_norm: plane output needs C %% 4 == 0 and 8-byte aligned planes");
    W2V2_REQUIRE(rows > 0 && C > 0 && C <= 2048, "layer_norm: rows=%lld C=%d unsupported (C <= 2048)",
                 (long long)rows, C);
    // 4 blocks per CU (16 waves), each wave looping over rows
    const int64_t want = (rows + 3) / 4;
    static int cap = -1;
    if (cap < 0) cap = tune_int("W2V2_LN_BLOCKS", 256 * 4);      // 512 / 1024 / 2048 / 4096 blocks -> 1.14 / 1.06 / 1.10 / 1.26 ms for the 25 LayerNorms of a base forward
    dim3 grid((unsigned)(want < cap ? want : cap)), block(256);
    ProfScope ps(prof, FAM_LAYERNORM, 8.0 * rows * C, (4.0 + (y ? 4.0 : 0.0) + (y16 ? 2.0 : 0.0) + (pl.p ? 2.0 * plane_count(pl.fmt) : 0.0)) * rows * C, s);
    // Round 4: plain rows (no activation, C % 4 == 0, C > 512, 16-byte aligned) take the row loop of the fused dropout kernel, whose next
    // row is loaded into the registers just consumed: 6.2 TB/s measured there against 4.8 for layer_norm_kernel's copy-forward prefetch.
    // Same element arithmetic in the same order: bit-identical.
    if (tune_int("W2V2_LN_V2", 1) != 0 && act == 0 && (C & 3) == 0 && C > 512 &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(y16) & 7) == 0) {
        if (C <= 1024)
            W2V2_LAUNCH((layer_norm_drop_kernel<4, false>), grid, block, 0, s, x, nullptr, nullptr, y, y16, gamma, beta, rows, C, eps, 0.f, (uint64_t)0, 0u, pl);
        else
            W2V2_LAUNCH((layer_norm_drop_kernel<8, false>), grid, block, 0, s, x, nullptr, nullptr, y, y16, gamma, beta, rows, C, eps, 0.f, (uint64_t)0, 0u, pl);
        W2V2_HIP_CHECK(hipGetLastError());
        return W2V2_OK;
    }
    if (C <= 256)
        W2V2_LAUNCH(layer_norm_kernel<1>, grid, block, 0, s, x, y, y16, gamma, beta, rows, C, eps, act, pl);
    else if (C <= 512)
        W2V2_LAUNCH(layer_norm_kernel<2>, grid, block, 0, s, x, y, y16, gamma, beta, rows, C, eps, act, pl);
    else if (C <= 1024)
        W2V2_LAUNCH(layer_norm_kernel<4>, grid, block, 0, s, x, y, y16, gamma, beta, rows, C, eps, act, pl);
    else
        W2V2_LAUNCH(layer_norm_kernel<8>, grid, block, 0, s, x, y, y16, gamma, beta, rows, C, eps, act, pl);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace w2v2
