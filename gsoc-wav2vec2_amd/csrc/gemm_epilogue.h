// Shared epilogue of the MFMA GEMM kernels: bias -> activation -> + residual -> store (fp32 and / or a bf16 shadow),
// from 32x32 accumulators in the C/D layout  col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
//
// A wave owns up to 64 outputs per lane, and with a short K (768 = 24 fp32 K tiles, 12 bf16 K tiles) the epilogue is
// a first-order cost -- on the bf16 kernel it used to be LONGER than the matrix work.  So: uniform conditions (which
// outputs exist, whether the sub-tile is interior) are decided once per wave, addresses are 32-bit offsets from a
// wave-uniform base (no 64-bit multiply-add per element), and exact GELU uses a 5-coefficient erf.
#pragma once

#include "common.h"
#include "train.h"

namespace w2v2 {

#ifdef __HIPCC__
using f32x16_t = __attribute__((ext_vector_type(16))) float;

// (gelu_erf_fast: common.h)
// C / C16 / R point at the wave sub-tile's first element (row 0, column 0 of the WTM x WTN block); any may be null.
// bias points at the sub-tile's first column.  rows_left / cols_left = valid extent of the sub-tile.
// c16_plane != 0: C16 receives the planes of the result, c16_plane elements apart: the exact three-term bf16 split (c16_fmt PF_BF16X3)
// or the two fp16 terms of result x F16X2_ACT_SCALE (PF_F16X2; a saturated value sets *range_flag).
template <int MT, int NTL, bool FAST_GELU>
__device__ __forceinline__ void gemm_epilogue(const f32x16_t (&acc)[MT][NTL], float* __restrict__ C, uint16_t* __restrict__ C16,
                                              const float* __restrict__ R, const float* __restrict__ bias, int ldc,
                                              int rows_left, int cols_left, int act, int li, int lh, int64_t c16_plane = 0, int c16_fmt = 0,
                                              int* range_flag = nullptr) {
    bool ovf = false;
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
                    for (int r = 0; r < 16; ++r) {
                        uint16_t* const d = C16 + off0 + ((r & 3) + 8 * (r >> 2)) * ldc;
                        if (c16_plane && c16_fmt == PF_F16X2) split2h_one(v[r], F16X2_ACT_SCALE, d[0], d[c16_plane], ovf);
                        else if (c16_plane) split3_one(v[r], d[0], d[c16_plane], d[2 * c16_plane]);
                        else d[0] = (uint16_t)pack_bf16_rne(v[r], 0.0f);
                    }
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
                        if (C16) {
                            if (c16_plane && c16_fmt == PF_F16X2) split2h_one(o, F16X2_ACT_SCALE, C16[off], C16[off + c16_plane], ovf);
                            else if (c16_plane) split3_one(o, C16[off], C16[off + c16_plane], C16[off + 2 * c16_plane]);
                            else C16[off] = (uint16_t)pack_bf16_rne(o, 0.0f);
                        }
                    }
                }
            }
        }
    }
    report_overflow(range_flag, ovf);
}

#endif

}  // namespace w2v2
