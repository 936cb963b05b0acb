// Entry points of the tools-only build (build.py --tuning, -DW2V2_TUNING -> lib/libw2v2_tuning.so): the per-block
// phase trace of the shadow-fed bf16 GEMM kernels (the op itself is w2v2_op_gemm_bf16_shadows of the product ABI).  Nothing here is compiled into
// the shipping library, and nothing on the product path calls it.
#ifdef W2V2_TUNING
#include "common.h"

namespace w2v2 {
extern unsigned long long* g_tune_trace;
}

extern "C" {

// 32 words per block of the NEXT launches of gemm_bf16_kernel: [0] = XCC_ID << 32 | HW_ID, [1] = wall clock (100 MHz) at entry,
// [2..] = shader clock at entry, first tile landed, after every k step, after the last MFMA block, stores retired; [31] = count
int w2v2_tune_set_trace(unsigned long long* dev_words) {
    w2v2::g_tune_trace = dev_words;
    return 0;
}

}  // extern "C"
#endif
