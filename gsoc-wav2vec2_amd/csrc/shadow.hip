// fp32 -> bf16 shadows for precision mode 1 (nearest-even, v_cvt_pk_bf16_f32): what gemm_bf16.hip would round an
// operand to on its own, materialised once so the GEMM streams 2 bytes per element instead of 4 and skips the
// conversion.  Activations get their shadow from the producing kernel (GEMM / LayerNorm / conv0 / attention
// epilogues); the kernels here serve the weights, which change only when variables are set or the optimizer steps.
#include <map>
#include <tuple>
#include <mutex>
#include <utility>

#include "common.h"

namespace w2v2 {
namespace {

__global__ __launch_bounds__(256) void to_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        *reinterpret_cast<uint2*>(y + i) = make_uint2(pack_bf16_rne(v.x, v.y), pack_bf16_rne(v.z, v.w));
    } else {
        for (int64_t j = i; j < n; ++j) y[j] = (uint16_t)pack_bf16_rne(x[j], 0.f);
    }
}

// w (K, N) row-major fp32  ->  wt (N, K) row-major bf16, through a 64 x 64 LDS tile (both sides coalesced)
__global__ __launch_bounds__(256) void transpose_to_bf16_kernel(const float* __restrict__ w, uint16_t* __restrict__ wt, int K, int N) {
    __shared__ float tile[64][65];
    w += (int64_t)blockIdx.z * K * N;            // batch of equally shaped matrices, densely packed on both sides
    wt += (int64_t)blockIdx.z * K * N;
    const int k0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int k = k0 + r, n = n0 + tx;
        tile[r][tx] = (k < K && n < N) ? w[(int64_t)k * N + n] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int n = n0 + r, k = k0 + tx;
        if (n < N && k < K) wt[(int64_t)n * K + k] = (uint16_t)pack_bf16_rne(tile[tx][r], 0.f);
    }
}

// All weight shadows of a model in ONE launch: block -> (weight, 64 x 64 tile) through a device table.  After every optimizer
// step the bf16 training path refreshes ~75 transposed shadows and ~55 plain copies; as separate launches that was ~130
// launches and ~1.3 ms per step for 0.75 GB of traffic.  A tile is read once (coalesced fp32), written as a plain bf16 copy
// (when the weight has one) and, through LDS, as the transposed (N, K) shadow.
__global__ __launch_bounds__(256) void weight_shadows_multi_kernel(const ShadowJob* __restrict__ jobs) {
    __shared__ float tile[64][65];
    const ShadowJob j = jobs[blockIdx.x];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    // whole tile inside, rows 16-byte aligned (every weight of the model shapes): 16-byte loads, 8-byte stores on both copies -- the element
    // loop below moved 4-byte loads and 2-byte stores (172 us per step for base, 553 for large-robust: 3.3 TB/s)
    const bool fast = j.k0 + 64 <= j.K && j.n0 + 64 <= j.N && (j.N & 3) == 0 && (j.K & 3) == 0 && (reinterpret_cast<uintptr_t>(j.w) & 15) == 0 &&
                      ((reinterpret_cast<uintptr_t>(j.plain) | reinterpret_cast<uintptr_t>(j.wt)) & 7) == 0;
    if (fast) {       // (block-uniform)
        const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;        // 16 lanes x float4 = one 64-wide row; 16 rows per pass
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = r0 + 16 * p;
            const int64_t src = (int64_t)(j.k0 + r) * j.N + j.n0 + c4;
            const float4 v = *reinterpret_cast<const float4*>(j.w + src);
            tile[r][c4] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
            if (j.plain) *reinterpret_cast<uint2*>(j.plain + src) = make_uint2(pack_bf16_rne(v.x, v.y), pack_bf16_rne(v.z, v.w));
        }
        __syncthreads();
        if (j.wt) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int n = r0 + 16 * p;                                    // output row (a column of the tile), 4 consecutive k per lane
                *reinterpret_cast<uint2*>(j.wt + (int64_t)(j.n0 + n) * j.K + j.k0 + c4) =
                    make_uint2(pack_bf16_rne(tile[c4][n], tile[c4 + 1][n]), pack_bf16_rne(tile[c4 + 2][n], tile[c4 + 3][n]));
            }
        }
        return;
    }
    for (int r = ty; r < 64; r += 4) {
        const int k = j.k0 + r, n = j.n0 + tx;
        const bool ok = k < j.K && n < j.N;
        const float v = ok ? j.w[(int64_t)k * j.N + n] : 0.f;
        tile[r][tx] = v;
        if (ok && j.plain) j.plain[(int64_t)k * j.N + n] = (uint16_t)pack_bf16_rne(v, 0.f);
    }
    __syncthreads();
    if (j.wt)
        for (int r = ty; r < 64; r += 4) {
            const int n = j.n0 + r, k = j.k0 + tx;
            if (n < j.N && k < j.K) j.wt[(int64_t)n * j.K + k] = (uint16_t)pack_bf16_rne(tile[tx][r], 0.f);
        }
}

// q | k | v packing of one layer's attention projections: three (H, H) kernels -> one (H, 3H), three (H) biases -> (3H)
// (and the reverse for their gradients).  One launch per layer instead of three strided + three linear copies.
struct QkvPtrs {
    float* packed_w;    // (H, 3H)
    float* packed_b;    // (3H)
    float* w[3];        // (H, H) each; null = skip (frozen variable on the unpack side)
    float* b[3];        // (H)
    int H;
};

template <bool PACK>
__global__ __launch_bounds__(256) void qkv_pack_kernel(QkvPtrs a) {
    const int j = blockIdx.y, H = a.H, hv = H >> 2;
    float* w = a.w[j];
    if (w) {
        const int64_t n4 = (int64_t)H * hv;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            const int64_t r = i / hv, c = (i % hv) * 4;
            float4* one = reinterpret_cast<float4*>(w + r * H + c);
            float4* all = reinterpret_cast<float4*>(a.packed_w + r * 3 * H + (int64_t)j * H + c);
            if (PACK) *all = *one; else *one = *all;
        }
    }
    float* b = a.b[j];
    if (b && blockIdx.x == 0)
        for (int i = threadIdx.x; i < H; i += 256) {
            if (PACK) a.packed_b[j * H + i] = b[i]; else b[i] = a.packed_b[j * H + i];
        }
}

// the same packing for up to QKV_MULTI layers in one launch (blockIdx.z = layer): the per-step refresh after the optimizer (w2v2_finalize)
constexpr int QKV_MULTI = 24;
struct QkvPtrsMulti {
    QkvPtrs l[QKV_MULTI];
};
__global__ __launch_bounds__(256) void qkv_pack_multi_kernel(QkvPtrsMulti t) {
    const QkvPtrs& a = t.l[blockIdx.z];
    const int j = blockIdx.y, H = a.H, hv = H >> 2;
    const float* w = a.w[j];
    const int64_t n4 = (int64_t)H * hv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / hv, c = (i % hv) * 4;
        *reinterpret_cast<float4*>(a.packed_w + r * 3 * H + (int64_t)j * H + c) = *reinterpret_cast<const float4*>(w + r * H + c);
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < H; i += 256) a.packed_b[j * H + i] = a.b[j][i];
}

}  // namespace

// layers' q | k | v kernels and biases -> packed (H, 3H) / (3H), QKV_MULTI layers per launch.  w / b: 3 pointers per layer.
int launch_qkv_pack_layers(float* const* packed_w, float* const* packed_b, const float* const* w, const float* const* b, int layers, int H, hipStream_t s) {
    W2V2_REQUIRE(packed_w && packed_b && w && b && layers > 0 && H > 0 && H % 4 == 0, "qkv_pack_layers: bad argument");
    const int blocks = (int)(((int64_t)H * H / 4 + 255) / 256);
    for (int l0 = 0; l0 < layers; l0 += QKV_MULTI) {
        const int n = layers - l0 < QKV_MULTI ? layers - l0 : QKV_MULTI;
        QkvPtrsMulti t;
        uintptr_t bits = 0;
        for (int i = 0; i < n; ++i) {
            QkvPtrs& a = t.l[i];
            a.packed_w = packed_w[l0 + i]; a.packed_b = packed_b[l0 + i]; a.H = H;
            bits |= reinterpret_cast<uintptr_t>(a.packed_w);
            for (int j = 0; j < 3; ++j) {
                a.w[j] = const_cast<float*>(w[3 * (l0 + i) + j]);
                a.b[j] = const_cast<float*>(b[3 * (l0 + i) + j]);
                W2V2_REQUIRE(a.w[j] && a.b[j] && a.packed_w && a.packed_b, "qkv_pack_layers: null source");
                bits |= reinterpret_cast<uintptr_t>(a.w[j]);
            }
        }
        for (int i = n; i < QKV_MULTI; ++i) t.l[i] = t.l[0];
        W2V2_REQUIRE((bits & 15) == 0, "qkv_pack_layers: unaligned buffer");
        W2V2_LAUNCH(qkv_pack_multi_kernel, dim3(blocks < 64 ? blocks : 64, 3, n), dim3(256), 0, s, t);
    }
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

static int launch_qkv(bool pack, float* packed_w, float* packed_b, float* const w[3], float* const b[3], int H, hipStream_t s) {
    W2V2_REQUIRE(packed_w && packed_b && H > 0 && H % 4 == 0, "qkv_pack: bad argument");
    QkvPtrs a;
    a.packed_w = packed_w; a.packed_b = packed_b; a.H = H;
    uintptr_t bits = reinterpret_cast<uintptr_t>(packed_w);
    for (int j = 0; j < 3; ++j) { a.w[j] = w[j]; a.b[j] = b[j]; bits |= reinterpret_cast<uintptr_t>(w[j]); }
    W2V2_REQUIRE((bits & 15) == 0, "qkv_pack: unaligned buffer");
    const int blocks = (int)(((int64_t)H * H / 4 + 255) / 256);
    if (pack) W2V2_LAUNCH(qkv_pack_kernel<true>, dim3(blocks < 256 ? blocks : 256, 3), dim3(256), 0, s, a);
    else W2V2_LAUNCH(qkv_pack_kernel<false>, dim3(blocks < 256 ? blocks : 256, 3), dim3(256), 0, s, a);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}
int launch_qkv_pack(float* packed_w, float* packed_b, const float* const w[3], const float* const b[3], int H, hipStream_t s) {
    float* wn[3] = {const_cast<float*>(w[0]), const_cast<float*>(w[1]), const_cast<float*>(w[2])};
    float* bn[3] = {const_cast<float*>(b[0]), const_cast<float*>(b[1]), const_cast<float*>(b[2])};
    W2V2_REQUIRE(wn[0] && wn[1] && wn[2] && bn[0] && bn[1] && bn[2], "qkv_pack: null source");
    return launch_qkv(true, packed_w, packed_b, wn, bn, H, s);
}
int launch_qkv_unpack(const float* packed_w, const float* packed_b, float* const w[3], float* const b[3], int H, hipStream_t s) {
    return launch_qkv(false, const_cast<float*>(packed_w), const_cast<float*>(packed_b), w, b, H, s);
}

// Grow-only device scratch owned by the library, one buffer per (purpose, device, stream): launches on one stream are
// ordered, so a buffer is never in use by two kernels at once, and different streams (other models, other host threads) get
// their own.  The device is part of the key because the null stream has the same handle on every device: one process
// driving two GPUs on their default streams must not be handed device 0's allocation while running on device 1.
namespace {
struct StreamScratch { void* p = nullptr; size_t bytes = 0; };
std::mutex g_scratch_mu;
std::map<std::tuple<int, int, hipStream_t>, StreamScratch> g_scratch;
}  // namespace

int stream_scratch(int slot, hipStream_t s, size_t bytes, void** out) {
    int dev = 0;
    W2V2_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    StreamScratch& e = g_scratch[std::make_tuple(slot, dev, s)];
    if (bytes > e.bytes) {
        if (e.p) W2V2_HIP_CHECK(hipFree(e.p));           // (hipFree waits for the device: no kernel still uses the old one)
        e.p = nullptr; e.bytes = 0;
        W2V2_HIP_CHECK(hipMalloc(&e.p, bytes));
        e.bytes = bytes;
    }
    *out = e.p;
    return W2V2_OK;
}

// Release every scratch buffer of the calling thread's current device (w2v2_release_scratch in the C ABI): the entries are
// otherwise kept for the life of the process.
int stream_scratch_release() {
    int dev = 0;
    W2V2_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    for (auto it = g_scratch.begin(); it != g_scratch.end();) {
        if (std::get<1>(it->first) == dev) {
            if (it->second.p) (void)hipFree(it->second.p);
            it = g_scratch.erase(it);
        } else {
            ++it;
        }
    }
    return W2V2_OK;
}

int launch_to_bf16(const float* x, uint16_t* y, int64_t n, hipStream_t s) {
    W2V2_REQUIRE(x && y && n > 0, "to_bf16: bad argument");
    W2V2_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0, "to_bf16: unaligned buffer");
    W2V2_LAUNCH(to_bf16_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, x, y, n);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_weight_shadows_multi(const ShadowJob* jobs_dev, int njobs, hipStream_t s) {
    W2V2_REQUIRE(jobs_dev && njobs > 0, "weight_shadows_multi: bad argument");
    ProfScope ps(tl_step_prof, FAM_OPTIMIZER, 0.0, 0.0, s);      // (the per-step refresh of the bf16 weight copies rides with the optimizer)
    W2V2_LAUNCH(weight_shadows_multi_kernel, dim3((unsigned)njobs), dim3(256), 0, s, jobs_dev);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_transpose_to_bf16(const float* w, uint16_t* wt, int K, int N, hipStream_t s) {
    return launch_transpose_to_bf16_batched(w, wt, K, N, 1, s);
}

int launch_transpose_to_bf16_batched(const float* w, uint16_t* wt, int K, int N, int nbatch, hipStream_t s) {
    W2V2_REQUIRE(w && wt && K > 0 && N > 0 && nbatch > 0, "transpose_to_bf16: bad argument");
    W2V2_LAUNCH(transpose_to_bf16_kernel, dim3((N + 63) / 64, (K + 63) / 64, nbatch), dim3(256), 0, s, w, wt, K, N);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace w2v2
