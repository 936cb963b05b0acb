// Shared epilogue of the MFMA GEMM kernels: bias -> activation -> + residual -> store (fp32 and / or a bf16 shadow),
// from 32x32 accumulators in the C/D layout  col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
//
// A wave owns up to 64 outputs per lane, and with a short K (768 = 24 fp32 K tiles, 12 bf16 K tiles) the epilogue is
// a first-order cost -- on the bf16 kernel it used to be LONGER than the matrix work.  So: uniform conditions (which
// outputs exist, whether the sub-tile is interior) are decided once per wave, addresses are 32-bit offsets from a
// wave-uniform base (no 64-bit multiply-add per element), and exact GELU uses a 5-coefficient erf.
#pragma once

#include "common.h"

namespace w2v2 {

#ifdef __HIPCC__
using f32x16_t = __attribute__((ext_vector_type(16))) float;

// exact GELU 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz-Stegun 7.1.26: |erf error| < 1.5e-7 absolute, ~14 VALU
// ops against ~35 for erff.  Used by the bf16 kernel only (FAST_GELU): its error is a smooth, i.e. BIASED, function of x,
// and through 19 GELU layers that bias moved the fp32 CTC loss of the 246000-sample fixture from 5e-3 to 1.7e-2 off the
// fp64 reference (logits 7.4e-5 -> 8.1e-5) -- invisible next to bf16 rounding, not acceptable for the fp32 path.
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
    const float erf_abs = fmaf(-p * t, e, 1.0f);             // erf(|x| / sqrt 2)
    return 0.5f * x + 0.5f * fabsf(x) * erf_abs;             // 0.5 x (1 + sign(x) erf(|x| / sqrt 2))
}

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
                for (int r = 0; r < 16; ++r) v[r] = FAST_GELU ? gelu_erf_fast(v[r]) : gelu_erf(v[r]);
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

// The same epilogue for accumulators held TRANSPOSED: the kernel issued mfma(B fragment, A fragment), so the 32x32 block
// in a lane's registers is C^T -- lane l owns output ROW (l & 31) and register r holds COLUMN (r & 3) + 8 (r >> 2) +
// 4 (l >> 5): four consecutive columns per register quad.  A lane therefore stores 16 bytes at a time (dwordx4 for the
// fp32 tensor, dwordx2 for its bf16 shadow, dwordx4 residual loads, float4 bias) -- 4 store instructions per accumulator
// instead of 16 (32 with the shadow).  With 16-deep bf16 MFMAs a K = 768 tile is only ~6 k matrix cycles per wave and
// the 128 narrow stores per lane of the row-major form were a store-issue tail longer than that; this form issues 32.
// Vector path: interior sub-tile, ldc % 4 == 0 and 16-byte (fp32) / 8-byte (bf16) aligned bases -- decided once per wave.
using f32x4_e = __attribute__((ext_vector_type(4))) float;
using u32x2_e = __attribute__((ext_vector_type(2))) unsigned;

template <int MT, int NTL, bool FAST_GELU>
__device__ __forceinline__ void gemm_epilogue_t(const f32x16_t (&acc)[MT][NTL], float* __restrict__ C, uint16_t* __restrict__ C16,
                                                const float* __restrict__ R, const float* __restrict__ bias, int ldc,
                                                int rows_left, int cols_left, int act, int li, int lh) {
    const bool interior = rows_left >= MT * 32 && cols_left >= NTL * 32;
    const bool vec = interior && (ldc & 3) == 0 && ((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(R)) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(C16) & 7) == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0;
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = nt * 32 + 8 * g + 4 * lh;                 // first of this lane's 4 consecutive columns
            f32x4_e bv = {0.f, 0.f, 0.f, 0.f};
            if (bias) {
                if (vec) {
                    bv = *reinterpret_cast<const f32x4_e*>(bias + c0);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bv[j] = (c0 + j < cols_left) ? bias[c0 + j] : 0.f;
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = mt * 32 + li;
                f32x4_e v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][4 * g + j] + bv[j];
                if (act == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = FAST_GELU ? gelu_erf_fast(v[j]) : gelu_erf(v[j]);
                } else if (act == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = gelu_tanh(v[j]);
                }
                const int off = row * ldc + c0;
                if (vec) {
                    if (R) v += *reinterpret_cast<const f32x4_e*>(R + off);
                    if (C) *reinterpret_cast<f32x4_e*>(C + off) = v;
                    if (C16) {
                        u32x2_e h;
                        h[0] = pack_bf16_rne(v[0], v[1]);
                        h[1] = pack_bf16_rne(v[2], v[3]);
                        *reinterpret_cast<u32x2_e*>(C16 + off) = h;
                    }
                } else if (row < rows_left) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (c0 + j < cols_left) {
                            float o = v[j];
                            if (R) o += R[off + j];
                            if (C) C[off + j] = o;
                            if (C16) C16[off + j] = (uint16_t)pack_bf16_rne(o, 0.0f);
                        }
                    }
                }
            }
        }
    }
}
#endif

}  // namespace w2v2
