// Row LayerNorm over the channel axis (+ optional GELU), fp32.
//
// tf.keras.layers.LayerNormalization(axis=-1, epsilon=eps) as the reference calls
// it (feature_extractor.py:48-50,86-88; encoder.py:96-98,105-108,232-234):
// population variance, y = (x - mean) * rsqrt(var + eps) * gamma + beta.
//
// HBM-bound: one read + one write of the row.  One wave64 per row; the row is
// held in registers (float4 per lane per 256-channel chunk) so mean and the
// centred variance are two in-register passes -- no E[x^2]-mean^2 cancellation.
#include "common.h"

namespace w2v2 {
namespace {

template <int NV>   // NV float4 per lane: covers C <= NV * 256
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* __restrict__ x,
                                                         float* __restrict__ y,
                                                         uint16_t* __restrict__ y16,   // optional bf16 shadow of y
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         int64_t rows, int C, float eps, int act) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    float* yr = y + row * C;
    const bool vec = (C & 3) == 0;
    float4 v[NV];
    float sum = 0.f;
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
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
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
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c >= C) continue;
        float o[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < C) o[e] = apply_act(o[e] * rstd * gamma[c + e] + beta[c + e], act);
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
        if (vec) {
            *reinterpret_cast<float4*>(yr + c) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < C) yr[c + e] = o[e];
        }
    }
}

}  // namespace

int launch_layer_norm(Profiler* prof, const float* x, float* y, const float* gamma,
                      const float* beta, int64_t rows, int C, float eps, int act, hipStream_t s) {
    return launch_layer_norm_x(prof, x, y, gamma, beta, rows, C, eps, act, nullptr, s);
}

int launch_layer_norm_x(Profiler* prof, const float* x, float* y, const float* gamma, const float* beta, int64_t rows,
                        int C, float eps, int act, uint16_t* y16, hipStream_t s) {
    W2V2_REQUIRE(x && y && gamma && beta, "layer_norm: null operand");
    W2V2_REQUIRE(rows > 0 && C > 0 && C <= 2048, "layer_norm: rows=%lld C=%d unsupported (C <= 2048)",
                 (long long)rows, C);
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    ProfScope ps(prof, FAM_LAYERNORM, 8.0 * rows * C, (y16 ? 10.0 : 8.0) * rows * C, s);
    if (C <= 256)
        hipLaunchKernelGGL(layer_norm_kernel<1>, grid, block, 0, s, x, y, y16, gamma, beta, rows, C, eps, act);
    else if (C <= 512)
        hipLaunchKernelGGL(layer_norm_kernel<2>, grid, block, 0, s, x, y, y16, gamma, beta, rows, C, eps, act);
    else if (C <= 1024)
        hipLaunchKernelGGL(layer_norm_kernel<4>, grid, block, 0, s, x, y, y16, gamma, beta, rows, C, eps, act);
    else
        hipLaunchKernelGGL(layer_norm_kernel<8>, grid, block, 0, s, x, y, y16, gamma, beta, rows, C, eps, act);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace w2v2
