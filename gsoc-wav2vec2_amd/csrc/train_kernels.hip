// Training-step support kernels (HBM-bound element-wise / reduction work around the MFMA kernels).
//
// Reference semantics: tf.keras.layers.Dropout (feature_extractor.py:90,95; encoder.py:20,42-44,94,118,
// 128,235,270; modeling.py:230,253): keep with probability 1-p, scale kept values by 1/(1-p).
// LayerNormalization / exact GELU backward are the analytic derivatives of the forward kernels.
// Keras Adam (main.py:211-216 -> tf.keras.optimizers.Adam defaults):
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
//
// Dropout masks are NOT stored: keep(seed, stream, index) is a counter-based hash (splitmix64), the same
// integer function as wav2vec2/variables.py::dropout_keep, regenerated wherever the mask is needed
// (forward, backward, and inside the attention kernels).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "train.h"

namespace w2v2 {
namespace {

constexpr int EW_THREADS = 256;

inline unsigned ew_grid(int64_t n_vec) {
    static int cap = -1;
    if (cap < 0) cap = tune_int("W2V2_EW_BLOCKS", 16384);      // 1024 / 2048 / 4096 / 16384 blocks -> 78.9 / 59.6 / 59.8 / 57.3 us per dropout_fwd launch (fine-tune step average)
    int64_t g = (n_vec + EW_THREADS - 1) / EW_THREADS;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));   // grid-stride beyond `cap` blocks
}

// Inputs kept as bf16 (precision mode 1: the FFN pre-activation u and the gradient of its activation are stored as bf16 only --
// EwBf16 in train.h): four consecutive elements starting at element index e, from the bf16 copy when there is one (8 bytes),
// else from the fp32 tensor (16 bytes), rounded to bf16 on the way in when `round_in` (the shadow-free path of the same mode).
__device__ __forceinline__ float bf16_round_trip(float v) { return __uint_as_float(pack_bf16_rne(v, 0.f) << 16); }
__device__ __forceinline__ void load4_maybe_bf16(const float* x, const uint16_t* x16, int64_t e, int round_in, float (&v)[4]) {
    if (x16) {
        const uint2 w = *reinterpret_cast<const uint2*>(x16 + e);
        v[0] = __uint_as_float(w.x << 16); v[1] = __uint_as_float(w.x & 0xFFFF0000u);
        v[2] = __uint_as_float(w.y << 16); v[3] = __uint_as_float(w.y & 0xFFFF0000u);
    } else {
        const float4 f = *reinterpret_cast<const float4*>(x + e);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
        if (round_in) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = bf16_round_trip(v[k]);
        }
    }
}
__device__ __forceinline__ float load1_maybe_bf16(const float* x, const uint16_t* x16, int64_t e, int round_in) {
    if (x16) return __uint_as_float((uint32_t)x16[e] << 16);
    return round_in ? bf16_round_trip(x[e]) : x[e];
}

// y = dropout(act(x)) [+ res]           (act may be 0).  HBM-bound: 16 bytes per lane per access when the tensor allows.
template <bool VEC>
__global__ void dropout_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                   float* __restrict__ y, uint16_t* __restrict__ y16 /* optional bf16 shadow of y */, int64_t n,
                                   int act, float p, uint64_t seed, uint32_t stream, const uint16_t* __restrict__ x16, int round_in) {
    const float inv = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const uint32_t key = dropout_key(seed, stream), thr = dropout_threshold(p);
    if (VEC) {
        const int64_t nv = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < nv; i += (int64_t)gridDim.x * EW_THREADS) {
            float xv[4], v[4];
            load4_maybe_bf16(x, x16, 4 * i, round_in, xv);
            if (act == 3) {
                const f32x2_t a0 = gelu_erf_fast2(f32x2_t{xv[0], xv[1]}), a1 = gelu_erf_fast2(f32x2_t{xv[2], xv[3]});
                v[0] = a0[0]; v[1] = a0[1]; v[2] = a1[0]; v[3] = a1[1];
            } else {
                v[0] = apply_act(xv[0], act); v[1] = apply_act(xv[1], act); v[2] = apply_act(xv[2], act); v[3] = apply_act(xv[3], act);
            }
            if (p > 0.f) {
#pragma unroll
                for (int e = 0; e < 4; e += 2) {       // elements 4 i + e, + 1 share one hash word
                    const uint32_t w = dropout_word(key, ((uint32_t)(4 * i) >> 1) + (e >> 1));
                    v[e] = dropout_keep_lo(w, thr) ? v[e] * inv : 0.0f;
                    v[e + 1] = dropout_keep_hi(w, thr) ? v[e + 1] * inv : 0.0f;
                }
            }
            if (res) {
                const float4 r = reinterpret_cast<const float4*>(res)[i];
                v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
            }
            if (y) reinterpret_cast<float4*>(y)[i] = make_float4(v[0], v[1], v[2], v[3]);
            if (y16) reinterpret_cast<uint2*>(y16)[i] = make_uint2(pack_bf16_rne(v[0], v[1]), pack_bf16_rne(v[2], v[3]));
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS) {
            float v = apply_act(load1_maybe_bf16(x, x16, i, round_in), act);
            if (p > 0.f) v = dropout_keep32(key, (uint32_t)i, thr) ? v * inv : 0.0f;
            v = res ? v + res[i] : v;
            if (y) y[i] = v;
            if (y16) y16[i] = (uint16_t)pack_bf16_rne(v, 0.f);
        }
    }
}

// The FFN's GELU + dropout on bf16-only tensors (u16 -> gd16; precision mode 1), eight elements per lane: 16-byte loads and stores instead of
// the general kernel's 8-byte ones on this path (a wave covers 1 KiB per access).  Same element arithmetic, pairing and hash indices as
// dropout_fwd_kernel<true>: bit-identical results.
__device__ __forceinline__ void unpack8_bf16(const uint4& w, float (&v)[8]) {
    v[0] = __uint_as_float(w.x << 16); v[1] = __uint_as_float(w.x & 0xFFFF0000u);
    v[2] = __uint_as_float(w.y << 16); v[3] = __uint_as_float(w.y & 0xFFFF0000u);
    v[4] = __uint_as_float(w.z << 16); v[5] = __uint_as_float(w.z & 0xFFFF0000u);
    v[6] = __uint_as_float(w.w << 16); v[7] = __uint_as_float(w.w & 0xFFFF0000u);
}
__global__ void dropout_fwd_bf16x8_kernel(const uint16_t* __restrict__ x16, uint16_t* __restrict__ y16, int64_t n8, int act, float p, uint64_t seed,
                                          uint32_t stream) {
    const float inv = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const uint32_t key = dropout_key(seed, stream), thr = dropout_threshold(p);
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n8; i += (int64_t)gridDim.x * EW_THREADS) {
        float xv[8], v[8];
        unpack8_bf16(reinterpret_cast<const uint4*>(x16)[i], xv);
        if (act == 3) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2_t a = gelu_erf_fast2(f32x2_t{xv[e], xv[e + 1]});
                v[e] = a[0]; v[e + 1] = a[1];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = apply_act(xv[e], act);
        }
        if (p > 0.f) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const uint32_t w = dropout_word(key, ((uint32_t)(8 * i) >> 1) + (e >> 1));
                v[e] = dropout_keep_lo(w, thr) ? v[e] * inv : 0.0f;
                v[e + 1] = dropout_keep_hi(w, thr) ? v[e + 1] * inv : 0.0f;
            }
        }
        reinterpret_cast<uint4*>(y16)[i] = make_uint4(pack_bf16_rne(v[0], v[1]), pack_bf16_rne(v[2], v[3]), pack_bf16_rne(v[4], v[5]), pack_bf16_rne(v[6], v[7]));
    }
}

// dx = dy * keep/(1-p) * act'(u)
// (dy16 may be dx16: every element is read once, then written, by the same lane)
template <bool VEC>
__global__ void dropout_bwd_kernel(const float* __restrict__ u, const float* __restrict__ dy,
                                   float* __restrict__ dx, uint16_t* dx16 /* optional bf16 shadow of dx */, int64_t n, int act,
                                   float p, uint64_t seed, uint32_t stream, const uint16_t* __restrict__ u16, const uint16_t* dy16, int round_in) {
    const float inv = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const uint32_t key = dropout_key(seed, stream), thr = dropout_threshold(p);
    if (VEC) {
        const int64_t nv = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < nv; i += (int64_t)gridDim.x * EW_THREADS) {
            float g[4];
            load4_maybe_bf16(dy, dy16, 4 * i, round_in, g);
            if (p > 0.f) {
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const uint32_t w = dropout_word(key, ((uint32_t)(4 * i) >> 1) + (e >> 1));
                    g[e] = dropout_keep_lo(w, thr) ? g[e] * inv : 0.0f;
                    g[e + 1] = dropout_keep_hi(w, thr) ? g[e + 1] * inv : 0.0f;
                }
            }
            if (act) {
                float uv[4];
                load4_maybe_bf16(u, u16, 4 * i, round_in, uv);
                if (act == 3) {
                    const f32x2_t d0 = gelu_grad_fast2(f32x2_t{uv[0], uv[1]}), d1 = gelu_grad_fast2(f32x2_t{uv[2], uv[3]});
                    g[0] *= d0[0]; g[1] *= d0[1]; g[2] *= d1[0]; g[3] *= d1[1];
                } else {
                    g[0] *= gelu_grad(uv[0], act); g[1] *= gelu_grad(uv[1], act); g[2] *= gelu_grad(uv[2], act); g[3] *= gelu_grad(uv[3], act);
                }
            }
            if (dx) reinterpret_cast<float4*>(dx)[i] = make_float4(g[0], g[1], g[2], g[3]);
            if (dx16) reinterpret_cast<uint2*>(dx16)[i] = make_uint2(pack_bf16_rne(g[0], g[1]), pack_bf16_rne(g[2], g[3]));
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS) {
            float g = load1_maybe_bf16(dy, dy16, i, round_in);
            if (p > 0.f) g = dropout_keep32(key, (uint32_t)i, thr) ? g * inv : 0.0f;
            if (act) g *= gelu_grad(load1_maybe_bf16(u, u16, i, round_in), act);
            if (dx) dx[i] = g;
            if (dx16) dx16[i] = (uint16_t)pack_bf16_rne(g, 0.f);
        }
    }
}

// dropout backward over a (rows, cols) tensor that ALSO leaves the column sums of its result: the result is the dY of a Dense
// layer, whose bias gradient is exactly that sum (fp32, unrounded).  Same element function and hash index (r cols + c) as
// dropout_bwd_kernel; the loop is the column-sum kernel's: 64 float4 column groups x 4 row lanes per block, 128-row chunks,
// partial[chunk][cols] folded by colsum_final_wide.  Needs cols % 4 == 0 and 16-byte aligned tensors.
__global__ __launch_bounds__(256) void dropout_bwd_colsum_kernel(const float* __restrict__ u, const float* __restrict__ dy,
                                                                 float* __restrict__ dx, uint16_t* dx16,
                                                                 float* __restrict__ partial, int64_t rows, int cols, int rows_per_chunk,
                                                                 int act, float p, uint64_t seed, uint32_t stream,
                                                                 const uint16_t* __restrict__ u16, const uint16_t* dy16, int round_in) {
    __shared__ float4 red[4][64];
    const float inv = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const uint32_t key = dropout_key(seed, stream), thr = dropout_threshold(p);
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + cx) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < cols) {
        for (int64_t r = r0 + ry; r < r1; r += 4) {
            const int64_t i = r * cols + c;
            float g[4];
            load4_maybe_bf16(dy, dy16, i, round_in, g);
            if (p > 0.f) {
#pragma unroll
                for (int e = 0; e < 4; e += 2) {       // i = r cols + c is a multiple of 4
                    const uint32_t w = dropout_word(key, ((uint32_t)i >> 1) + (e >> 1));
                    g[e] = dropout_keep_lo(w, thr) ? g[e] * inv : 0.0f;
                    g[e + 1] = dropout_keep_hi(w, thr) ? g[e + 1] * inv : 0.0f;
                }
            }
            if (act) {
                float uv[4];
                load4_maybe_bf16(u, u16, i, round_in, uv);
                if (act == 3) {
                    const f32x2_t d0 = gelu_grad_fast2(f32x2_t{uv[0], uv[1]}), d1 = gelu_grad_fast2(f32x2_t{uv[2], uv[3]});
                    g[0] *= d0[0]; g[1] *= d0[1]; g[2] *= d1[0]; g[3] *= d1[1];
                } else {
                    g[0] *= gelu_grad(uv[0], act); g[1] *= gelu_grad(uv[1], act); g[2] *= gelu_grad(uv[2], act); g[3] *= gelu_grad(uv[3], act);
                }
            }
            if (dx) *reinterpret_cast<float4*>(dx + i) = make_float4(g[0], g[1], g[2], g[3]);
            if (dx16) *reinterpret_cast<uint2*>(dx16 + i) = make_uint2(pack_bf16_rne(g[0], g[1]), pack_bf16_rne(g[2], g[3]));
            acc.x += g[0]; acc.y += g[1]; acc.z += g[2]; acc.w += g[3];
        }
    }
    red[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && c < cols) {
        const float4 p0 = red[0][cx], p1 = red[1][cx], p2 = red[2][cx], p3 = red[3][cx];
        *reinterpret_cast<float4*>(partial + (int64_t)blockIdx.y * cols + c) =
            make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
                        (p0.w + p1.w) + (p2.w + p3.w));
    }
}

// dropout_bwd_colsum_kernel on bf16-only tensors (u16, dy16 -> dx16, in place or not), eight columns per lane: 16-byte accesses; a block is
// 32 column groups (256 columns) x 8 row lanes.  Same element arithmetic and hash indices; the column sums add the rows in a different
// order than the four-column kernel (fp32 summation order only).
__global__ __launch_bounds__(256) void dropout_bwd_colsum_bf16x8_kernel(const uint16_t* __restrict__ u16, const uint16_t* dy16, uint16_t* dx16,
                                                                        float* __restrict__ partial, int64_t rows, int cols, int rows_per_chunk,
                                                                        int act, float p, uint64_t seed, uint32_t stream) {
    __shared__ float red[8][32][9];
    const float inv = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const uint32_t key = dropout_key(seed, stream), thr = dropout_threshold(p);
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = (blockIdx.x * 32 + cx) * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (c < cols) {
        for (int64_t r = r0 + ry; r < r1; r += 8) {
            const int64_t i = r * cols + c;                       // a multiple of 8
            float g[8];
            unpack8_bf16(*reinterpret_cast<const uint4*>(dy16 + i), g);
            if (p > 0.f) {
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const uint32_t w = dropout_word(key, ((uint32_t)i >> 1) + (e >> 1));
                    g[e] = dropout_keep_lo(w, thr) ? g[e] * inv : 0.0f;
                    g[e + 1] = dropout_keep_hi(w, thr) ? g[e + 1] * inv : 0.0f;
                }
            }
            if (act) {
                float uv[8];
                unpack8_bf16(*reinterpret_cast<const uint4*>(u16 + i), uv);
                if (act == 3) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2_t d = gelu_grad_fast2(f32x2_t{uv[e], uv[e + 1]});
                        g[e] *= d[0]; g[e + 1] *= d[1];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] *= gelu_grad(uv[e], act);
                }
            }
            *reinterpret_cast<uint4*>(dx16 + i) = make_uint4(pack_bf16_rne(g[0], g[1]), pack_bf16_rne(g[2], g[3]), pack_bf16_rne(g[4], g[5]), pack_bf16_rne(g[6], g[7]));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += g[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ry][cx][e] = acc[e];
    __syncthreads();
    {   // 256 threads fold the 8 row lanes of the block's 256 columns: thread = one column
        const int col = threadIdx.x, gx = col >> 3, ge = col & 7;
        if (blockIdx.x * 256 + col < cols) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += red[j][gx][ge];
            partial[(int64_t)blockIdx.y * cols + blockIdx.x * 256 + col] = t;
        }
    }
}

// dst (Kin, Nout) = A16^T B16 over a FEW rows (the rows a weight gradient's fast slabs leave over): bf16 operands (exact
// products), fp32 accumulation in row order.  Thread = 4 consecutive output columns n x 8 consecutive output rows k; the A run
// of a row is block-uniform (scalar loads), the B run is one 8-byte load.
__global__ __launch_bounds__(256) void dw_tail_bf16_kernel(const uint16_t* __restrict__ A16, int64_t lda, const uint16_t* __restrict__ B16,
                                                           int64_t ldb, float* __restrict__ dst, int R, int Kin, int Nout) {
    const int n = (blockIdx.x * 256 + threadIdx.x) * 4, k0 = blockIdx.y * 8;
    if (n >= Nout) return;
    float acc[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
#pragma unroll 4
    for (int r = 0; r < R; ++r) {
        const uint2 bv = *reinterpret_cast<const uint2*>(B16 + (int64_t)r * ldb + n);
        const float b[4] = {__uint_as_float(bv.x << 16), __uint_as_float(bv.x & 0xFFFF0000u), __uint_as_float(bv.y << 16),
                            __uint_as_float(bv.y & 0xFFFF0000u)};
        const uint4 av = *reinterpret_cast<const uint4*>(A16 + (int64_t)r * lda + k0);       // block-uniform: 8 bf16
        const uint32_t aw[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a0 = __uint_as_float(aw[j] << 16), a1 = __uint_as_float(aw[j] & 0xFFFF0000u);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * j][e] = fmaf(a0, b[e], acc[2 * j][e]);
                acc[2 * j + 1][e] = fmaf(a1, b[e], acc[2 * j + 1][e]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
        *reinterpret_cast<float4*>(dst + (int64_t)(k0 + j) * Nout + n) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
}

// batched transpose: y[b][c][r] = x[b][r][c]
__global__ void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int cols) {
    __shared__ float tile[32][33];
    const int64_t boff = (int64_t)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int r = r0 + ty + j, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + j][tx] = x[boff + (int64_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int c = c0 + ty + j, r = r0 + tx;
        if (r < rows && c < cols) y[boff + (int64_t)c * rows + r] = tile[tx][ty + j];
    }
}

// stage 1: partial[chunk][c] = sum over the chunk's rows of x[r][c]   (fp32 within a chunk).  HBM-bound, so the launch must
// put thousands of loads in flight: VEC blocks are 64 float4 column groups (256 columns) x 4 row lanes, every thread keeps
// 4 rows in flight, and chunks are 128 rows -- (cols / 256) x (rows / 128) blocks (576 for the (24576, 768) tensors; the first
// version's 1024-column blocks gave those 96 blocks for 256 CUs and ran at 1.9 TB/s).
template <bool VEC>
__global__ void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                      int64_t rows, int cols, int rows_per_chunk) {
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
    if (VEC) {
        __shared__ float4 red[4][64];
        const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
        const int c = (blockIdx.x * 64 + cx) * 4;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        if (c < cols) {
            int64_t r = r0 + ry;
            for (; r + 12 < r1; r += 16) {
                const float4 v0 = *reinterpret_cast<const float4*>(x + r * cols + c);
                const float4 v1 = *reinterpret_cast<const float4*>(x + (r + 4) * cols + c);
                const float4 v2 = *reinterpret_cast<const float4*>(x + (r + 8) * cols + c);
                const float4 v3 = *reinterpret_cast<const float4*>(x + (r + 12) * cols + c);
                a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
                a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
                a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
                a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
            }
            for (; r < r1; r += 4) {
                const float4 v0 = *reinterpret_cast<const float4*>(x + r * cols + c);
                a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
            }
        }
        red[ry][cx] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                                  (a0.w + a1.w) + (a2.w + a3.w));
        __syncthreads();
        if (ry == 0 && c < cols) {
            const float4 p0 = red[0][cx], p1 = red[1][cx], p2 = red[2][cx], p3 = red[3][cx];
            *reinterpret_cast<float4*>(partial + (int64_t)blockIdx.y * cols + c) =
                make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
                            (p0.w + p1.w) + (p2.w + p3.w));
        }
    } else {
        const int c = blockIdx.x * EW_THREADS + threadIdx.x;
        if (c >= cols) return;
        float acc = 0.f;
        for (int64_t r = r0; r < r1; ++r) acc += x[r * cols + c];
        partial[(int64_t)blockIdx.y * cols + c] = acc;
    }
}
// stage 2: out[c] (+)= sum over chunks, fp64 accumulate
template <bool VEC>
__global__ void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                    int nchunks, int cols, int64_t ld, int accumulate) {
    if (VEC) {
        const int c = (blockIdx.x * EW_THREADS + threadIdx.x) * 4;
        if (c >= cols) return;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int k = 0; k < nchunks; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(partial + (int64_t)k * ld + c);
            a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
        }
        float4 o = make_float4((float)a0, (float)a1, (float)a2, (float)a3);
        if (accumulate) {
            const float4 p = *reinterpret_cast<const float4*>(out + c);
            o = make_float4(p.x + o.x, p.y + o.y, p.z + o.z, p.w + o.w);
        }
        *reinterpret_cast<float4*>(out + c) = o;
    } else {
        const int c = blockIdx.x * EW_THREADS + threadIdx.x;
        if (c >= cols) return;
        double acc = 0.0;
        for (int k = 0; k < nchunks; ++k) acc += (double)partial[(int64_t)k * ld + c];
        out[c] = accumulate ? out[c] + (float)acc : (float)acc;
    }
}
// stage 2 for many chunks x few columns: 32 columns per block, the chunk loop split over 8 thread rows
// (blockIdx.y = g selects one of up to three column groups that share the partial rows: partial + g cols -> out / out1 / out2)
__global__ __launch_bounds__(256) void colsum_final_wide_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                int nchunks, int cols, int64_t ld, int accumulate,
                                                                float* __restrict__ out1 = nullptr, float* __restrict__ out2 = nullptr) {
    __shared__ double red[8][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    partial += (int64_t)blockIdx.y * cols;
    out = blockIdx.y == 0 ? out : (blockIdx.y == 1 ? out1 : out2);
    double acc = 0.0;
    if (c < cols) {
        int k = ry;
        for (; k + 24 < nchunks; k += 32) {          // four independent loads in flight per lane
            const float v0 = partial[(int64_t)k * ld + c], v1 = partial[(int64_t)(k + 8) * ld + c];
            const float v2 = partial[(int64_t)(k + 16) * ld + c], v3 = partial[(int64_t)(k + 24) * ld + c];
            acc += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; k < nchunks; k += 8) acc += (double)partial[(int64_t)k * ld + c];
    }
    red[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && c < cols) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += red[j][cx];
        out[c] = accumulate ? out[c] + (float)t : (float)t;
    }
}

// The deferred folds of one encoder layer in one launch (train.h: FoldBatch).  A block finds its job by a wave-uniform scan of the
// table (at most FOLD_MAX_JOBS entries, in the kernel-argument segment) and then runs the body of the kernel the job replaces, with the
// same per-column summation order.
__global__ __launch_bounds__(256) void fold_multi_kernel(const FoldTable tab) {
    __shared__ double red[8][33];
    int ji = 0;
    while (ji + 1 < tab.n && (int)blockIdx.x >= tab.j[ji + 1].block0) ++ji;
    const FoldJob& J = tab.j[ji];
    const int b = (int)blockIdx.x - J.block0;
    const float* __restrict__ partial = J.partial;
    const int nchunks = J.nchunks, cols = J.cols;
    const int64_t ld = J.ld;
    if (J.kind == 0) {      // colsum_final_kernel<true>
        const int c = (b * EW_THREADS + (int)threadIdx.x) * 4;
        if (c >= cols) return;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int k = 0;
        for (; k + 4 <= nchunks; k += 4) {        // four slab loads in flight, added in slab order (the same sums as the one-by-one loop)
            const float4 v0 = *reinterpret_cast<const float4*>(partial + (int64_t)k * ld + c);
            const float4 v1 = *reinterpret_cast<const float4*>(partial + (int64_t)(k + 1) * ld + c);
            const float4 v2 = *reinterpret_cast<const float4*>(partial + (int64_t)(k + 2) * ld + c);
            const float4 v3 = *reinterpret_cast<const float4*>(partial + (int64_t)(k + 3) * ld + c);
            a0 += (double)v0.x; a1 += (double)v0.y; a2 += (double)v0.z; a3 += (double)v0.w;
            a0 += (double)v1.x; a1 += (double)v1.y; a2 += (double)v1.z; a3 += (double)v1.w;
            a0 += (double)v2.x; a1 += (double)v2.y; a2 += (double)v2.z; a3 += (double)v2.w;
            a0 += (double)v3.x; a1 += (double)v3.y; a2 += (double)v3.z; a3 += (double)v3.w;
        }
        for (; k < nchunks; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(partial + (int64_t)k * ld + c);
            a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
        }
        const float4 o = make_float4((float)a0, (float)a1, (float)a2, (float)a3);
        const int H = J.unpack;
        if (H) {            // column c of the packed (H, 3H) matrix = row c / 3H, kernel (c % 3H) / H, column c % H  (H % 4 == 0: a float4 stays in one kernel)
            const int row = c / (3 * H), rem = c - row * 3 * H, which = rem / H;
            float* dst = which == 0 ? J.out[0] : (which == 1 ? J.out[1] : J.out[2]);
            *reinterpret_cast<float4*>(dst + (int64_t)row * H + (rem - which * H)) = o;
        } else {
            *reinterpret_cast<float4*>(J.out[0] + c) = o;
        }
        return;
    }
    // colsum_final_wide_kernel: blockIdx = (column block, group)
    const int colblocks = (cols + 31) / 32;
    const int g = b / colblocks, bx = b - g * colblocks;
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = bx * 32 + cx;
    partial += (int64_t)g * cols;
    float* out = g == 0 ? J.out[0] : (g == 1 ? J.out[1] : J.out[2]);
    double acc = 0.0;
    if (c < cols) {
        int k = ry;
        for (; k + 24 < nchunks; k += 32) {
            const float v0 = partial[(int64_t)k * ld + c], v1 = partial[(int64_t)(k + 8) * ld + c];
            const float v2 = partial[(int64_t)(k + 16) * ld + c], v3 = partial[(int64_t)(k + 24) * ld + c];
            acc += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; k < nchunks; k += 8) acc += (double)partial[(int64_t)k * ld + c];
    }
    red[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && c < cols) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += red[j][cx];
        out[c] = (float)t;
    }
}

// LayerNorm backward.  One wave per row (grid-stride); per-lane dgamma / dbeta partials live in
// registers across the rows a wave visits, are combined across the block's 4 waves through LDS and
// written as partial[block][2][C]; colsum_final reduces them.
// DXSUM: also leave the column sums of dx (dx is the dY of the Dense layer in front of this LayerNorm: its bias gradient) as a
// third group of C partials.
// TAIL 2: dx is (plus the residual) the gradient of t1 = dropout(o) + x of the attention block: the dropout backward of the
// out-projection's output rides along -- dx16 then receives bf16(dropout-backward(dx)) (the dY shadow of that Dense layer) and the
// third partial group its column sums (the out-projection's bias gradient); dx itself is still written in fp32.
template <int NV, int TAIL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ dy, float* __restrict__ dx,
                                                     uint16_t* __restrict__ dx16 /* optional bf16 shadow of dx (TAIL 2: of its dropout backward) */,
                                                     float* __restrict__ partial, int64_t rows, int C, float eps,
                                                     const float* __restrict__ res /* optional: dx = LN-backward + res */,
                                                     float drop_p, uint64_t drop_seed, uint32_t drop_stream) {
    extern __shared__ float red[];   // 4 waves x NG x C
    constexpr bool DXSUM = TAIL != 0;
    constexpr int NG = DXSUM ? 3 : 2;
    const uint32_t dkey = TAIL == 2 ? dropout_key(drop_seed, drop_stream) : 0u, dthr = TAIL == 2 ? dropout_threshold(drop_p) : 0u;
    const float dinv = (TAIL == 2 && drop_p > 0.f) ? 1.0f / (1.0f - drop_p) : 1.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[NV][4], db[NV][4], ds[DXSUM ? NV : 1][4];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) dg[i][e] = db[i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < (DXSUM ? NV : 1); ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) ds[i][e] = 0.f;
    const bool vec = (C & 3) == 0;
    float gam[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (i * 64 + lane) * 4 + e;
            gam[i][e] = c < C ? gamma[c] : 0.f;
        }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const float* xr = x + row * C;
        const float* dr = dy + row * C;
        float xv[NV][4], gv[NV][4], dv[NV][4];
        float sum = 0.f;
        if (vec) {      // 16-byte loads of x and dy issued together
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                const float4 a = c < C ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 d = c < C ? *reinterpret_cast<const float4*>(dr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                xv[i][0] = a.x; xv[i][1] = a.y; xv[i][2] = a.z; xv[i][3] = a.w;
                dv[i][0] = d.x; dv[i][1] = d.y; dv[i][2] = d.z; dv[i][3] = d.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = (i * 64 + lane) * 4 + e;
                    xv[i][e] = c < C ? xr[c] : 0.f;
                    dv[i][e] = c < C ? dr[c] : 0.f;
                }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += xv[i][e];
        const float mean = wave_sum(sum) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (i * 64 + lane) * 4 + e;
                xv[i][e] = c < C ? xv[i][e] - mean : 0.f;
                sq += xv[i][e] * xv[i][e];
            }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)C + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (i * 64 + lane) * 4 + e;
                const float d = dv[i][e];
                const float xh = xv[i][e] * rstd;
                xv[i][e] = xh;
                gv[i][e] = c < C ? d * gam[i][e] : 0.f;
                dg[i][e] += d * xh;
                db[i][e] += d;
                s1 += gv[i][e];
                s2 += gv[i][e] * xh;
            }
        s1 = wave_sum(s1) / (float)C;
        s2 = wave_sum(s2) / (float)C;
        float* dxr = dx + row * C;
        if (vec) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < C) {
                    float4 o = make_float4(rstd * (gv[i][0] - s1 - xv[i][0] * s2), rstd * (gv[i][1] - s1 - xv[i][1] * s2),
                                           rstd * (gv[i][2] - s1 - xv[i][2] * s2), rstd * (gv[i][3] - s1 - xv[i][3] * s2));
                    if (res) {
                        const float4 rv = *reinterpret_cast<const float4*>(res + row * C + c);
                        o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
                    }
                    *reinterpret_cast<float4*>(dxr + c) = o;
                    if constexpr (TAIL == 2) {        // o -> dropout backward of o (the same pair-wise decisions as dropout_bwd_kernel)
                        if (drop_p > 0.f) {
                            const uint32_t pr = (uint32_t)(row * C + c) >> 1;
                            const uint32_t w0 = dropout_word(dkey, pr), w1 = dropout_word(dkey, pr + 1);
                            o.x = dropout_keep_lo(w0, dthr) ? o.x * dinv : 0.f;
                            o.y = dropout_keep_hi(w0, dthr) ? o.y * dinv : 0.f;
                            o.z = dropout_keep_lo(w1, dthr) ? o.z * dinv : 0.f;
                            o.w = dropout_keep_hi(w1, dthr) ? o.w * dinv : 0.f;
                        }
                    }
                    if (dx16) *reinterpret_cast<uint2*>(dx16 + row * C + c) = make_uint2(pack_bf16_rne(o.x, o.y), pack_bf16_rne(o.z, o.w));
                    if constexpr (DXSUM) { ds[i][0] += o.x; ds[i][1] += o.y; ds[i][2] += o.z; ds[i][3] += o.w; }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = (i * 64 + lane) * 4 + e;
                    if (c < C) {
                        float o = rstd * (gv[i][e] - s1 - xv[i][e] * s2);
                        if (res) o += res[row * C + c];
                        dxr[c] = o;
                        if constexpr (TAIL == 2) {
                            if (drop_p > 0.f) o = dropout_keep32(dkey, (uint32_t)(row * C + c), dthr) ? o * dinv : 0.f;
                        }
                        if (dx16) dx16[row * C + c] = (uint16_t)pack_bf16_rne(o, 0.f);
                        if constexpr (DXSUM) ds[i][e] += o;
                    }
                }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (i * 64 + lane) * 4 + e;
            if (c < C) {
                red[(wave * NG + 0) * C + c] = dg[i][e];
                red[(wave * NG + 1) * C + c] = db[i][e];
                if constexpr (DXSUM) red[(wave * NG + 2) * C + c] = ds[i][e];
            }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < NG * C; c += 256) {
        const int which = c / C, cc = c % C;
        partial[(int64_t)blockIdx.x * NG * C + c] =
            red[(0 * NG + which) * C + cc] + red[(1 * NG + which) * C + cc] + red[(2 * NG + which) * C + cc] + red[(3 * NG + which) * C + cc];
    }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr_t, float b1, float b2, float eps) {
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// All trainable variables in ONE launch: block -> (variable, 4096-element chunk) through a small table.  The 200-odd
// per-variable launches of the first version cost 1.2 ms of launch latency per step for 0.3 ms of memory traffic.
__global__ void adam_multi_kernel(const AdamChunk* __restrict__ chunks, const float* __restrict__ grads, float* __restrict__ am,
                                  float* __restrict__ av, float lr_t, float b1, float b2, float eps) {
    const AdamChunk c = chunks[blockIdx.x];
    float* __restrict__ p = c.p;
    const float* __restrict__ g = grads + c.goff;
    float* __restrict__ m = am + c.goff;
    float* __restrict__ v = av + c.goff;
    for (int i = threadIdx.x; i < c.n; i += EW_THREADS) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// y[row] = mask[row] ? embed : x[row]      (spec_augment.py:127, tf.where(mask, spec_embed, x))
__global__ void spec_aug_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                    const float* __restrict__ embed, float* __restrict__ y, int64_t rows, int H) {
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < rows * H; i += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t r = i / H;
        y[i] = mask[r] ? embed[i % H] : x[i];
    }
}
// dx[row] = mask[row] ? 0 : dy[row];   masked rows' dy are kept in dmasked (then column-summed into d embed)
__global__ void spec_aug_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ mask,
                                    float* __restrict__ dx, float* __restrict__ dmasked, int64_t rows, int H) {
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < rows * H; i += (int64_t)gridDim.x * EW_THREADS) {
        const bool mk = mask[i / H] != 0;
        const float g = dy[i];
        dx[i] = mk ? 0.f : g;
        dmasked[i] = mk ? g : 0.f;
    }
}

__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                             int64_t n, float alpha, float beta) {
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS)
        y[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}
// 16 bytes per lane, optional bf16 shadow of the result (n % 4 == 0, aligned operands)
__global__ void axpby4_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, uint16_t* __restrict__ y16,
                              int64_t n4, float alpha, float beta) {
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_THREADS) {
        const float4 av = reinterpret_cast<const float4*>(a)[i];
        float4 o = make_float4(alpha * av.x, alpha * av.y, alpha * av.z, alpha * av.w);
        if (b) {
            const float4 bv = reinterpret_cast<const float4*>(b)[i];
            o.x += beta * bv.x; o.y += beta * bv.y; o.z += beta * bv.z; o.w += beta * bv.w;
        }
        reinterpret_cast<float4*>(y)[i] = o;
        if (y16) reinterpret_cast<uint2*>(y16)[i] = make_uint2(pack_bf16_rne(o.x, o.y), pack_bf16_rne(o.z, o.w));
    }
}

// zero rows t >= frame_len[b] of a (B, T, H) tensor (encoder.py:253 and its gradient)
__global__ void mask_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ frame_len,
                                 float* __restrict__ y, int B, int T, int H) {
    const int64_t n = (int64_t)B * T * H;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t row = i / H;
        const int b = (int)(row / T), t = (int)(row % T);
        y[i] = t < frame_len[b] ? x[i] : 0.f;
    }
}

}  // namespace

int launch_dropout_fwd(const float* x, const float* res, float* y, int64_t n, int act, float p,
                       uint64_t seed, uint32_t stream_id, hipStream_t s) {
    return launch_dropout_fwd_x(x, res, y, nullptr, n, act, p, seed, stream_id, s);
}

int launch_dropout_fwd_x(const float* x, const float* res, float* y, uint16_t* y16, int64_t n, int act, float p,
                         uint64_t seed, uint32_t stream_id, hipStream_t s, const EwBf16& in) {
    W2V2_REQUIRE((x || in.a16) && (y || y16) && n > 0 && p >= 0.f && p < 1.f, "dropout_fwd: bad argument");      // (y may be null: bf16 result only)
    ProfScope ps(tl_step_prof, FAM_DROPOUT, 0.0, (double)n * ((x ? 4.0 : 2.0) + (res ? 4.0 : 0.0) + (y ? 4.0 : 0.0) + (y16 ? 2.0 : 0.0)), s);
    const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res)) & 15) == 0 &&
                     ((reinterpret_cast<uintptr_t>(y16) | reinterpret_cast<uintptr_t>(in.a16)) & 7) == 0;
    // bf16 in, bf16 out, nothing else (the FFN's GELU + dropout in precision mode 1): eight elements per lane
    const bool x8 = in.a16 && !res && !y && y16 && (n & 7) == 0 && ((reinterpret_cast<uintptr_t>(y16) | reinterpret_cast<uintptr_t>(in.a16)) & 15) == 0 &&
                    tune_int("W2V2_EW_X8", 1) != 0;
    if (x8)
        W2V2_LAUNCH(dropout_fwd_bf16x8_kernel, dim3(ew_grid(n >> 3)), dim3(EW_THREADS), 0, s, in.a16, y16, n >> 3, act, p, seed, stream_id);
    else if (vec)
        W2V2_LAUNCH(dropout_fwd_kernel<true>, dim3(ew_grid(n >> 2)), dim3(EW_THREADS), 0, s, x, res, y, y16, n, act, p, seed, stream_id, in.a16,
                           in.round_in);
    else
        W2V2_LAUNCH(dropout_fwd_kernel<false>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, s, x, res, y, y16, n, act, p, seed, stream_id, in.a16,
                           in.round_in);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_dropout_bwd(const float* u, const float* dy, float* dx, int64_t n, int act, float p,
                       uint64_t seed, uint32_t stream_id, hipStream_t s) {
    return launch_dropout_bwd_x(u, dy, dx, nullptr, n, act, p, seed, stream_id, s);
}

int launch_dropout_bwd_x(const float* u, const float* dy, float* dx, uint16_t* dx16, int64_t n, int act, float p,
                         uint64_t seed, uint32_t stream_id, hipStream_t s, const EwBf16& in) {
    W2V2_REQUIRE((dy || in.b16) && (dx || dx16) && n > 0 && p >= 0.f && p < 1.f && (act == 0 || u || in.a16), "dropout_bwd: bad argument");   // (dx may be null)
    ProfScope ps(tl_step_prof, FAM_DROPOUT, 0.0, (double)n * ((dy ? 4.0 : 2.0) + (act ? (u ? 4.0 : 2.0) : 0.0) + (dx ? 4.0 : 0.0) + (dx16 ? 2.0 : 0.0)), s);
    const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0 &&
                     ((reinterpret_cast<uintptr_t>(dx16) | reinterpret_cast<uintptr_t>(in.a16) | reinterpret_cast<uintptr_t>(in.b16)) & 7) == 0;
    if (vec)
        W2V2_LAUNCH(dropout_bwd_kernel<true>, dim3(ew_grid(n >> 2)), dim3(EW_THREADS), 0, s, u, dy, dx, dx16, n, act, p, seed, stream_id, in.a16,
                           in.b16, in.round_in);
    else
        W2V2_LAUNCH(dropout_bwd_kernel<false>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, s, u, dy, dx, dx16, n, act, p, seed, stream_id, in.a16,
                           in.b16, in.round_in);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_dw_tail_bf16(const uint16_t* A16, int64_t lda, const uint16_t* B16, int64_t ldb, float* dst, int R, int Kin, int Nout, hipStream_t s) {
    W2V2_REQUIRE(A16 && B16 && dst && R > 0 && Kin > 0 && Nout > 0 && Kin % 8 == 0 && lda % 8 == 0 && Nout % 4 == 0 && ldb % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(A16) & 15) == 0 && (reinterpret_cast<uintptr_t>(B16) & 7) == 0 &&
                     (reinterpret_cast<uintptr_t>(dst) & 15) == 0,
                 "dw_tail_bf16: bad argument");
    ProfScope ps(tl_step_prof, FAM_GEMM_BF16, 2.0 * R * (double)Kin * Nout, 2.0 * R * ((double)Kin + Nout) + 4.0 * (double)Kin * Nout, s);
    W2V2_LAUNCH(dw_tail_bf16_kernel, dim3((Nout / 4 + 255) / 256, Kin / 8), dim3(256), 0, s, A16, lda, B16, ldb, dst, R, Kin, Nout);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_transpose(const float* x, float* y, int rows, int cols, int nbatch, hipStream_t s) {
    W2V2_REQUIRE(x && y && rows > 0 && cols > 0 && nbatch > 0, "transpose: bad argument");
    ProfScope ps(tl_step_prof, FAM_MISC, 0.0, 8.0 * rows * (double)cols * nbatch, s);
    dim3 grid((cols + 31) / 32, (rows + 31) / 32, nbatch);
    W2V2_LAUNCH(transpose_kernel, grid, dim3(256), 0, s, x, y, rows, cols);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

constexpr int COLSUM_CHUNK = 128;    // rows per stage-1 block

int64_t colsum_ws_floats(int64_t rows, int cols) { return ((rows + COLSUM_CHUNK - 1) / COLSUM_CHUNK) * (int64_t)cols + 8; }
// launch_dropout_bwd_colsum may cut its chunks down to 16 rows
int64_t dropout_bwd_colsum_ws_floats(int64_t rows, int cols) { return ((rows + 15) / 16) * (int64_t)cols + 8; }

// out[c] = sum over `nrows` rows of already-reduced partial rows (a producer's per-block column sums): one launch, fp64 accumulate
int launch_colsum_fold(const float* partial, float* out, int nrows, int cols, hipStream_t s) {
    W2V2_REQUIRE(partial && out && nrows > 0 && cols > 0, "colsum_fold: bad argument");
    ProfScope ps(tl_step_prof, FAM_REDUCE, 0.0, 4.0 * (nrows + 1.0) * cols, s);
    W2V2_LAUNCH(colsum_final_wide_kernel, dim3((cols + 31) / 32), dim3(256), 0, s, partial, out, nrows, cols, (int64_t)cols, 0, nullptr, nullptr);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

bool FoldBatch::add_tall(const float* partial, int nchunks, int64_t cols, float* out, int unpack, float* out1, float* out2) {
    if (tab.n >= FOLD_MAX_JOBS || !partial || !out || nchunks <= 0 || cols <= 0 || cols > 0x7FFFFFF0 || (cols & 3) != 0) return false;
    uintptr_t al = reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(out);
    if (unpack) {
        if (!out1 || !out2 || (unpack & 3) != 0 || cols != (int64_t)3 * unpack * unpack) return false;
        al |= reinterpret_cast<uintptr_t>(out1) | reinterpret_cast<uintptr_t>(out2);
    }
    if (al & 15) return false;
    FoldJob& J = tab.j[tab.n++];
    J.partial = partial; J.out[0] = out; J.out[1] = out1; J.out[2] = out2;
    J.ld = cols; J.nchunks = nchunks; J.cols = (int)cols; J.kind = 0; J.groups = 1; J.unpack = unpack; J.block0 = nblocks;
    nblocks += (int)((cols + 4 * EW_THREADS - 1) / (4 * EW_THREADS));
    bytes += 4.0 * (nchunks + 1.0) * (double)cols;
    return true;
}

bool FoldBatch::add_wide(const float* partial, int nchunks, int cols, int64_t ld, int groups, float* out0, float* out1, float* out2) {
    if (tab.n >= FOLD_MAX_JOBS || !partial || nchunks <= 0 || cols <= 0 || groups < 1 || groups > 3) return false;
    if (!out0 || (groups > 1 && !out1) || (groups > 2 && !out2)) return false;
    FoldJob& J = tab.j[tab.n++];
    J.partial = partial; J.out[0] = out0; J.out[1] = out1; J.out[2] = out2;
    J.ld = ld; J.nchunks = nchunks; J.cols = cols; J.kind = 1; J.groups = groups; J.unpack = 0; J.block0 = nblocks;
    nblocks += ((cols + 31) / 32) * groups;
    bytes += 4.0 * (nchunks + 1.0) * (double)cols * groups;
    return true;
}

int FoldBatch::flush(hipStream_t s) {
    if (tab.n == 0) return W2V2_OK;
    {
        ProfScope ps(tl_step_prof, FAM_REDUCE, 0.0, bytes, s);
        W2V2_LAUNCH(fold_multi_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, tab);
    }
    tab.n = 0; nblocks = 0; bytes = 0.0;
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_colsum(const float* x, float* out, int64_t rows, int cols, float* ws, int accumulate, hipStream_t s) {
    W2V2_REQUIRE(x && out && rows > 0 && cols > 0, "colsum: bad argument");
    ProfScope ps(tl_step_prof, FAM_REDUCE, 0.0, 4.0 * (rows + 1.0) * cols, s);
    const bool vec = (cols & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(ws)) & 15) == 0;
    const int per_block = vec ? 4 * EW_THREADS : EW_THREADS;
    if (rows <= 64) {   // few rows (split-K slabs): one pass, fp64 accumulate, no scratch
        if (vec)
            W2V2_LAUNCH(colsum_final_kernel<true>, dim3((cols + per_block - 1) / per_block), dim3(EW_THREADS), 0, s, x, out,
                               (int)rows, cols, (int64_t)cols, accumulate);
        else
            W2V2_LAUNCH(colsum_final_kernel<false>, dim3((cols + per_block - 1) / per_block), dim3(EW_THREADS), 0, s, x, out,
                               (int)rows, cols, (int64_t)cols, accumulate);
        W2V2_HIP_CHECK(hipGetLastError());
        return W2V2_OK;
    }
    W2V2_REQUIRE(ws, "colsum: null workspace");
    const int nchunks = (int)((rows + COLSUM_CHUNK - 1) / COLSUM_CHUNK);
    dim3 grid(vec ? (cols + 255) / 256 : (cols + per_block - 1) / per_block, nchunks);
    if (vec)
        W2V2_LAUNCH(colsum_partial_kernel<true>, grid, dim3(EW_THREADS), 0, s, x, ws, rows, cols, COLSUM_CHUNK);
    else
        W2V2_LAUNCH(colsum_partial_kernel<false>, grid, dim3(EW_THREADS), 0, s, x, ws, rows, cols, COLSUM_CHUNK);
    W2V2_LAUNCH(colsum_final_wide_kernel, dim3((cols + 31) / 32), dim3(256), 0, s, ws, out, nchunks, cols, (int64_t)cols, accumulate);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

// dx = dropout-backward(dy) as launch_dropout_bwd_x, plus colsum[c] = sum over rows of dx[r][c] (the bias gradient of the Dense
// layer dx is the output gradient of).  ws: dropout_bwd_colsum_ws_floats(rows, cols) floats.  Falls back to the two separate passes when
// the tensors do not allow 16-byte accesses.
int launch_dropout_bwd_colsum(const float* u, const float* dy, float* dx, uint16_t* dx16, float* colsum, int64_t rows, int cols,
                              int act, float p, uint64_t seed, uint32_t stream_id, float* ws, hipStream_t s, const EwBf16& in, FoldBatch* defer) {
    W2V2_REQUIRE((dy || in.b16) && (dx || dx16) && colsum && ws && rows > 0 && cols > 0 && p >= 0.f && p < 1.f && (act == 0 || u || in.a16),
                 "dropout_bwd_colsum: bad argument");
    ProfScope ps(tl_step_prof, FAM_DROPOUT, 0.0,
                 (double)rows * cols * ((dy ? 4.0 : 2.0) + (act ? (u ? 4.0 : 2.0) : 0.0) + (dx ? 4.0 : 0.0) + (dx16 ? 2.0 : 0.0)), s);
    const bool vec = (cols & 3) == 0 && ((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) |
                                           reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(colsum)) & 15) == 0 &&
                     ((reinterpret_cast<uintptr_t>(dx16) | reinterpret_cast<uintptr_t>(in.a16) | reinterpret_cast<uintptr_t>(in.b16)) & 7) == 0;
    if (!vec) {
        W2V2_REQUIRE(dx, "dropout_bwd_colsum: a bf16-only result needs cols %% 4 == 0 and 16-byte aligned tensors");
        if (int e = launch_dropout_bwd_x(u, dy, dx, dx16, rows * cols, act, p, seed, stream_id, s, in)) return e;
        return launch_colsum(dx, colsum, rows, cols, ws, 0, s);
    }
    // shorter chunks until the grid has `target` blocks (this HBM-bound kernel wants many small blocks; the fold of the partial
    // rows grows with them);
    // ws: dropout_bwd_colsum_ws_floats(rows, cols) floats
    static int target = -1;
    if (target < 0) target = tune_int("W2V2_DBC_BLOCKS", 8192);      // >= 2048 / 4096 / 8192 / 16384 blocks -> 20.4 / 18.8 / 17.8 / 18.9 ms (kernel + fold, 7 fine-tune steps)
    int chunk = COLSUM_CHUNK;
    const int colblocks = (cols + 255) / 256;
    while (chunk > 16 && (rows + chunk - 1) / chunk * colblocks < target) chunk >>= 1;
    const int nchunks = (int)((rows + chunk - 1) / chunk);
    // bf16 operands only (u16, dy16 -> dx16): eight columns per lane
    const bool x8 = !dx && dx16 && in.b16 && (!act || in.a16) && (cols & 7) == 0 && tune_int("W2V2_EW_X8", 1) != 0 &&
                    ((reinterpret_cast<uintptr_t>(dx16) | reinterpret_cast<uintptr_t>(in.a16) | reinterpret_cast<uintptr_t>(in.b16)) & 15) == 0;
    if (x8)
        W2V2_LAUNCH(dropout_bwd_colsum_bf16x8_kernel, dim3(colblocks, nchunks), dim3(256), 0, s, in.a16, in.b16, dx16, ws, rows, cols, chunk, act, p, seed,
                           stream_id);
    else
        W2V2_LAUNCH(dropout_bwd_colsum_kernel, dim3(colblocks, nchunks), dim3(256), 0, s, u, dy, dx, dx16, ws, rows, cols,
                           chunk, act, p, seed, stream_id, in.a16, in.b16, in.round_in);
    if (!(defer && defer->add_wide(ws, nchunks, cols, (int64_t)cols, 1, colsum)))
        W2V2_LAUNCH(colsum_final_wide_kernel, dim3((cols + 31) / 32), dim3(256), 0, s, ws, colsum, nchunks, cols, (int64_t)cols, 0);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

static int ln_bwd_blocks(int64_t rows) {
    // two blocks per CU: measured at 24576 x 768 -- 1024 blocks 45.8 / 49.4 us (without / with the dx column sums) + 12 us for
    // the fold of twice as many partial rows, 512 blocks 41.0 / 47.2 + 7, 256 blocks 62 / 67
    int64_t b = (rows + 3) / 4;
    return (int)(b > 512 ? 512 : b);
}
int64_t ln_bwd_ws_floats(int64_t rows, int C) { return (int64_t)ln_bwd_blocks(rows) * 3 * C + 8; }

int launch_ln_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma,
                  float* dbeta, int64_t rows, int C, float eps, float* ws, hipStream_t s) {
    return launch_ln_bwd_x(x, gamma, dy, dx, nullptr, dgamma, dbeta, rows, C, eps, ws, s, nullptr, nullptr, nullptr, nullptr);
}

int launch_ln_bwd_x(const float* x, const float* gamma, const float* dy, float* dx, uint16_t* dx16, float* dgamma,
                    float* dbeta, int64_t rows, int C, float eps, float* ws, hipStream_t s, float* dxsum, const float* residual,
                    const LnDropTail* tail, FoldBatch* defer) {
    W2V2_REQUIRE(!tail || (dx16 && dxsum && tail->p >= 0.f && tail->p < 1.f), "ln_bwd: the dropout tail needs its bf16 output and a column-sum target");
    W2V2_REQUIRE(x && gamma && dy && dx && dgamma && dbeta && ws, "ln_bwd: null operand");
    W2V2_REQUIRE(!residual || (C & 3) != 0 || (reinterpret_cast<uintptr_t>(residual) & 15) == 0, "ln_bwd: unaligned residual");
    W2V2_REQUIRE(!dx16 || ((C & 3) == 0 && (reinterpret_cast<uintptr_t>(dx16) & 7) == 0), "ln_bwd: the bf16 shadow needs C %% 4 == 0");
    W2V2_REQUIRE(rows > 0 && C > 0 && C <= 1024, "ln_bwd: rows=%lld C=%d unsupported (C <= 1024)", (long long)rows, C);
    ProfScope ps(tl_step_prof, FAM_LN_BWD, 0.0, (double)rows * C * (12.0 + (residual ? 4.0 : 0.0) + (dx16 ? 2.0 : 0.0)), s);
    const int nb = ln_bwd_blocks(rows);
    const int ng = dxsum ? 3 : 2;
    const size_t lds = (size_t)4 * ng * C * sizeof(float);
    auto go = [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        if (tail)
            W2V2_LAUNCH((ln_bwd_kernel<NV, 2>), dim3(nb), dim3(256), lds, s, x, gamma, dy, dx, dx16, ws, rows, C, eps, residual, tail->p,
                               tail->seed, tail->stream);
        else if (dxsum)
            W2V2_LAUNCH((ln_bwd_kernel<NV, 1>), dim3(nb), dim3(256), lds, s, x, gamma, dy, dx, dx16, ws, rows, C, eps, residual, 0.f,
                               (uint64_t)0, 0u);
        else
            W2V2_LAUNCH((ln_bwd_kernel<NV, 0>), dim3(nb), dim3(256), lds, s, x, gamma, dy, dx, dx16, ws, rows, C, eps, residual, 0.f,
                               (uint64_t)0, 0u);
    };
    if (C <= 256) go(std::integral_constant<int, 1>{});
    else if (C <= 512) go(std::integral_constant<int, 2>{});
    else go(std::integral_constant<int, 4>{});
    // partial is (nb, ng C): dgamma = column sums of its first C columns, dbeta of the next C [, the sums of dx of the last C]
    if (!(defer && defer->add_wide(ws, nb, C, (int64_t)ng * C, ng, dgamma, dbeta, dxsum)))
        W2V2_LAUNCH(colsum_final_wide_kernel, dim3((C + 31) / 32, ng), dim3(256), 0, s, ws, dgamma, nb, C, (int64_t)ng * C, 0, dbeta, dxsum);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float b1, float b2,
                float eps, hipStream_t s) {
    W2V2_REQUIRE(p && g && m && v && n > 0, "adam: bad argument");
    ProfScope ps(tl_step_prof, FAM_OPTIMIZER, 0.0, 28.0 * n, s);
    W2V2_LAUNCH(adam_kernel, dim3(ew_grid(n)), dim3(EW_THREADS), 0, s, p, g, m, v, n, lr_t, b1, b2, eps);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_adam_multi(const AdamChunk* chunks_dev, int nchunks, const float* grads, float* m, float* v, float lr_t, float b1,
                      float b2, float eps, hipStream_t s) {
    W2V2_REQUIRE(chunks_dev && nchunks > 0 && grads && m && v, "adam_multi: bad argument");
    ProfScope ps(tl_step_prof, FAM_OPTIMIZER, 0.0, 28.0 * 4096.0 * nchunks, s);
    W2V2_LAUNCH(adam_multi_kernel, dim3(nchunks), dim3(EW_THREADS), 0, s, chunks_dev, grads, m, v, lr_t, b1, b2, eps);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_spec_aug_fwd(const float* x, const uint8_t* mask, const float* embed, float* y, int64_t rows, int H, hipStream_t s) {
    W2V2_REQUIRE(x && mask && embed && y && rows > 0 && H > 0, "spec_aug_fwd: bad argument");
    ProfScope ps(tl_step_prof, FAM_MISC, 0.0, 8.0 * rows * H, s);
    W2V2_LAUNCH(spec_aug_fwd_kernel, dim3(ew_grid(rows * H)), dim3(EW_THREADS), 0, s, x, mask, embed, y, rows, H);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_spec_aug_bwd(const float* dy, const uint8_t* mask, float* dx, float* dmasked, int64_t rows, int H, hipStream_t s) {
    W2V2_REQUIRE(dy && mask && dx && dmasked && rows > 0 && H > 0, "spec_aug_bwd: bad argument");
    ProfScope ps(tl_step_prof, FAM_MISC, 0.0, 8.0 * rows * H, s);
    W2V2_LAUNCH(spec_aug_bwd_kernel, dim3(ew_grid(rows * H)), dim3(EW_THREADS), 0, s, dy, mask, dx, dmasked, rows, H);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_axpby(const float* a, const float* b, float* y, int64_t n, float alpha, float beta, hipStream_t s) {
    return launch_axpby_x(a, b, y, nullptr, n, alpha, beta, s);
}

// y = alpha a + beta b (b may be null); y16: optional bf16 shadow of y (written only on the 16-byte path; returns an error if
// it was asked for and the operands do not allow that path, so a caller never consumes a shadow that was not written)
int launch_axpby_x(const float* a, const float* b, float* y, uint16_t* y16, int64_t n, float alpha, float beta, hipStream_t s) {
    W2V2_REQUIRE(a && y && n > 0, "axpby: bad argument");
    ProfScope ps(tl_step_prof, FAM_MISC, 0.0, (double)n * (8.0 + (b ? 4.0 : 0.0) + (y16 ? 2.0 : 0.0)), s);
    const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(y16) & 7) == 0;
    if (vec) {
        W2V2_LAUNCH(axpby4_kernel, dim3(ew_grid(n >> 2)), dim3(EW_THREADS), 0, s, a, b, y, y16, n >> 2, alpha, beta);
        W2V2_HIP_CHECK(hipGetLastError());
        return W2V2_OK;
    }
    W2V2_REQUIRE(!y16, "axpby: the bf16 shadow needs n %% 4 == 0 and 16-byte aligned operands");
    W2V2_LAUNCH(axpby_kernel, dim3(ew_grid(n)), dim3(EW_THREADS), 0, s, a, b, y, n, alpha, beta);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_mask_rows(const float* x, const int32_t* frame_len, float* y, int B, int T, int H, hipStream_t s) {
    W2V2_REQUIRE(x && frame_len && y && B > 0 && T > 0 && H > 0, "mask_rows: bad argument");
    ProfScope ps(tl_step_prof, FAM_MISC, 0.0, 8.0 * B * (double)T * H, s);
    W2V2_LAUNCH(mask_rows_kernel, dim3(ew_grid((int64_t)B * T * H)), dim3(EW_THREADS), 0, s, x, frame_len, y, B, T, H);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace w2v2
