// Entry points of the tools-only build (build.py --tuning, -DW2V2_TUNING -> lib/libw2v2_tuning.so): op-level access to the
// shadow-fed bf16 GEMM with bf16 operands handed in directly, and the per-block phase trace.  Nothing here is compiled into
// the shipping library, and nothing on the product path calls it.
#ifdef W2V2_TUNING
#include "common.h"

namespace w2v2 {
extern unsigned long long* g_tune_trace;
}

extern "C" {

// C (+ C16) = act(A16 . B16^T + bias) + residual with A16 (M, K) bf16 rows lda apart and B16 the (N, K) bf16 shadow of the weight
int w2v2_tune_gemm16(const uint16_t* A16, int64_t lda, int64_t strideA, const uint16_t* B16, float* C, uint16_t* C16, int64_t ldc,
                     int64_t strideC, const float* bias, const float* residual, int32_t M, int32_t N, int32_t K, int32_t nbatch,
                     int32_t act, void* stream) {
    w2v2::GemmShadows x;
    x.A16 = A16; x.B16 = B16; x.C16 = C16; x.ldb16 = K;
    return w2v2::launch_gemm_bf16_x(nullptr, nullptr, lda, strideA, nullptr, N, 0, C, ldc, strideC, bias, residual, M, N, K, nbatch, act, x,
                                    reinterpret_cast<hipStream_t>(stream));
}

// 32 words per block of the NEXT launches of gemm_bf16_kernel: [0] = XCC_ID << 32 | HW_ID, [1] = wall clock (100 MHz) at entry,
// [2..] = shader clock at entry, first tile landed, after every k step, after the last MFMA block, stores retired; [31] = count
int w2v2_tune_set_trace(unsigned long long* dev_words) {
    w2v2::g_tune_trace = dev_words;
    return 0;
}

}  // extern "C"
#endif
