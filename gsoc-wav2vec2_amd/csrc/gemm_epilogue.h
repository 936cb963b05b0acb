// Shared epilogue of the MFMA GEMM kernels: bias -> activation -> + residual -> store (fp32 and / or a bf16 shadow),
// from 32x32 accumulators in the C/D layout  col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
//
// A wave owns up to 64 outputs per lane, and with a short K (768 = 24 fp32 K tiles, 12 bf16 K tiles) the epilogue is
// a first-order cost -- on the bf16 kernel it used to be LONGER than the matrix work.  So: uniform conditions (which
// outputs exist, whether the sub-tile is interior) are decided once per wave, addresses are 32-bit offsets from a
// wave-uniform base (no 64-bit multiply-add per element), and exact GELU uses a 5-coefficient erf.
#pragma once

#include <utility>

#include "common.h"
#include "train.h"

namespace w2v2 {

#ifdef __HIPCC__
using f32x16_t = __attribute__((ext_vector_type(16))) float;

// (gelu_erf_fast: common.h)
// C / C16 / R point at the wave sub-tile's first element (row 0, column 0 of the WTM x WTN block); any may be null.
// bias points at the sub-tile's first column.  rows_left / cols_left = valid extent of the sub-tile.
template <int MT, int NTL, bool FAST_GELU>
__device__ __forceinline__ void gemm_epilogue(const f32x16_t (&acc)[MT][NTL], float* __restrict__ C, uint16_t* __restrict__ C16,
                                              const float* __restrict__ R, const float* __restrict__ bias, int ldc,
                                              int rows_left, int cols_left, int act, int li, int lh) {
    const bool interior = rows_left >= MT * 32 && cols_left >= NTL * 32;
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
        const int cl = nt * 32 + li;                          // column inside the wave sub-tile
        const bool col_ok = cl < cols_left;
        const float bv = (bias && col_ok) ? bias[cl] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x16_t v = acc[mt][nt];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += bv;
            if (act == 1) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    if (FAST_GELU) {
                        const f32x2_t g = gelu_erf_fast2(f32x2_t{v[r], v[r + 1]});
                        v[r] = g[0];
                        v[r + 1] = g[1];
                    } else {
                        v[r] = gelu_erf(v[r]);
                        v[r + 1] = gelu_erf(v[r + 1]);
                    }
                }
            } else if (act == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = gelu_tanh(v[r]);
            }
            const int off0 = (mt * 32 + 4 * lh) * ldc + cl;   // register r adds ((r & 3) + 8 (r >> 2)) * ldc
            if (interior) {
                if (R) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += R[off0 + ((r & 3) + 8 * (r >> 2)) * ldc];
                }
                if (C) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) C[off0 + ((r & 3) + 8 * (r >> 2)) * ldc] = v[r];
                }
                if (C16) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) C16[off0 + ((r & 3) + 8 * (r >> 2)) * ldc] = (uint16_t)pack_bf16_rne(v[r], 0.0f);
                }
            } else if (col_ok) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (rl < rows_left) {
                        const int off = off0 + ((r & 3) + 8 * (r >> 2)) * ldc;
                        float o = v[r];
                        if (R) o += R[off];
                        if (C) C[off] = o;
                        if (C16) C16[off] = (uint16_t)pack_bf16_rne(o, 0.0f);
                    }
                }
            }
        }
    }
}

// ---- training epilogues of the bf16 GEMM (precision mode 1) ------------------------------------------------------------------
// The element-wise kernels either side of a Dense layer of the fine-tune step are pure HBM traffic over the GEMM's own output
// (an F-wide one moves 600 MB per layer); here they ride in the epilogue, on the accumulators:
//   MODE 1 (forward):   u = acc + bias  [-> pre, fp32]  -> act -> dropout -> + residual -> C / C16
//                       (FFN up-projection: pre = u for the backward, C16 = dropout(GELU(u));  attention out-projection:
//                        C = dropout(o) + x)
//   MODE 2 (backward):  g = dropout-backward(acc) * act'(u) -> C / C16, and this wave's column sums of g -> colpart
//                       (dY of the FFN up-projection from the down-projection's data-gradient GEMM; the sums are its bias gradient)
// Same element function, hash index (row * ldc + col, so ldc must be the tensor's row length) and operation order as
// dropout_fwd_kernel / dropout_bwd_kernel (train_kernels.hip): fused and unfused results are bitwise equal, the column sums equal up
// to fp32 summation order.  One hash word serves the elements (row, 2j) and (row, 2j + 1), which sit in neighbouring LANES of the
// C/D layout: the even lane hashes the rows of the even registers, the odd lane those of the odd registers, one DPP swap each.
// The keep decision is applied as an integer mask (the compare / select form costs 32 lane masks: attention_bf16.hip).
struct GemmTrainEpiDev {
    int mode;                  // 0 = none (plain gemm_epilogue)
    int act;                   // 0 | 1 | 2 | 3 (apply_act / gelu_grad numbering)
    float inv;                 // 1 / (1 - p)
    uint32_t key, thr1;        // dropout_key(seed, stream), dropout_threshold(p) - 1 (wraps to 0xFFFFFFFF for p = 0: keep all)
    float* pre;                // MODE 1, optional
    const float* u;            // MODE 2 (act != 0)
    float* colpart;            // MODE 2, optional: (ceil(M / wave-tile rows)) x N
};

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>).  The element function below is ~100
// instructions per accumulator block; left as a `#pragma unroll` loop over MT = 4 blocks hipcc declines to unroll it and indexes the
// accumulators through scratch (576 B per lane in the 128 x 256 kernel's instances).
template <class F, int... I>
__device__ __forceinline__ void static_for_seq(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_seq(f, std::make_integer_sequence<int, N>{});
}

template <int MT, int NTL, int MODE>
__device__ __forceinline__ void gemm_epilogue_train(const f32x16_t (&acc)[MT][NTL], float* __restrict__ C, uint16_t* __restrict__ C16,
                                                    const float* __restrict__ R, const float* __restrict__ bias, float* __restrict__ pre,
                                                    const float* __restrict__ U, float* __restrict__ colpart /* this wave's row, or null */,
                                                    int ldc, int rows_left, int cols_left, int act, uint32_t pair0 /* flat index of the
                                                    sub-tile's first element >> 1 */, uint32_t key, uint32_t thr1, float inv, int li, int lh) {
    const bool interior = rows_left >= MT * 32 && cols_left >= NTL * 32;
    const uint32_t odd = (uint32_t)li & 1u, hp = (uint32_t)ldc >> 1;
    static_for<NTL>([&](auto NTc) {
        constexpr int nt = decltype(NTc)::value;
        const int cl = nt * 32 + li;
        const bool col_ok = cl < cols_left;
        const float bv = (MODE == 1 && bias && col_ok) ? bias[cl] : 0.0f;
        float csum = 0.0f;
        static_for<MT>([&](auto MTc) {
            constexpr int mt = decltype(MTc)::value;
            f32x16_t v = acc[mt][nt];
            const int off0 = (mt * 32 + 4 * lh) * ldc + cl;   // register r adds ((r & 3) + 8 (r >> 2)) * ldc
            bool ok[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) ok[r] = interior || (col_ok && mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh < rows_left);
            if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += bv;
                if (pre) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (ok[r]) pre[off0 + ((r & 3) + 8 * (r >> 2)) * ldc] = v[r];
                }
                if (act == 1 || act == 3) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        if (act == 3) {
                            const f32x2_t g = gelu_erf_fast2(f32x2_t{v[r], v[r + 1]});
                            v[r] = g[0];
                            v[r + 1] = g[1];
                        } else {
                            v[r] = gelu_erf(v[r]);
                            v[r + 1] = gelu_erf(v[r + 1]);
                        }
                    }
                } else if (act == 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = gelu_tanh(v[r]);
                }
            }
            // dropout (forward and backward are the same map): keep ? v / (1 - p) : 0
            // (pair index = pbase + row offset * hp: pre-multiplied once, the per-register part is a small multiple of hp * FIB)
            const uint32_t pbase = pair0 + (uint32_t)(mt * 32 + 4 * lh) * hp + ((uint32_t)cl >> 1);
            const uint32_t pm0 = (pbase + odd * hp) * DROPOUT_FIB, hpm = hp * DROPOUT_FIB;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                // this lane hashes the row of register r + odd: row offset (r & 3) + odd + 8 (r >> 2)   (r even: no carry into bit 2)
                const uint32_t w_self = dropout_word_premul(key, pm0 + (uint32_t)((r & 3) + 8 * (r >> 2)) * hpm);
                const uint32_t w_peer = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w_self, 0xB1 /* quad_perm [1, 0, 3, 2] */, 0xF, 0xF, false);
                const uint32_t w0 = odd ? w_peer : w_self, w1 = odd ? w_self : w_peer;     // rows of registers r, r + 1
                const uint32_t h0 = odd ? (w0 >> 16) : (w0 & 0xFFFFu), h1 = odd ? (w1 >> 16) : (w1 & 0xFFFFu);
                const uint32_t m0 = (uint32_t)((int32_t)(thr1 - h0) >> 31), m1 = (uint32_t)((int32_t)(thr1 - h1) >> 31);
                v[r] = __uint_as_float(__float_as_uint(v[r] * inv) & m0);
                v[r + 1] = __uint_as_float(__float_as_uint(v[r + 1] * inv) & m1);
            }
            if (MODE == 2 && act) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float u0 = ok[r] ? U[off0 + ((r & 3) + 8 * (r >> 2)) * ldc] : 0.0f;
                    const float u1 = ok[r + 1] ? U[off0 + (((r + 1) & 3) + 8 * ((r + 1) >> 2)) * ldc] : 0.0f;
                    if (act == 3) {
                        const f32x2_t d = gelu_grad_fast2(f32x2_t{u0, u1});
                        v[r] *= d[0];
                        v[r + 1] *= d[1];
                    } else {
                        v[r] *= gelu_grad(u0, act);
                        v[r + 1] *= gelu_grad(u1, act);
                    }
                }
            }
            if (MODE == 1 && R) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (ok[r]) v[r] += R[off0 + ((r & 3) + 8 * (r >> 2)) * ldc];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (ok[r]) {
                    const int off = off0 + ((r & 3) + 8 * (r >> 2)) * ldc;
                    if (C) C[off] = v[r];
                    if (C16) C16[off] = (uint16_t)pack_bf16_rne(v[r], 0.0f);
                    if (MODE == 2) csum += v[r];
                }
            }
        });
        if (MODE == 2 && colpart) {
            csum += __shfl_xor(csum, 32);
            if (lh == 0 && col_ok) colpart[cl] = csum;
        }
    });
}

#endif

}  // namespace w2v2
