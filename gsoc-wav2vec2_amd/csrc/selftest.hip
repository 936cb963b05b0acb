// Device self-checks exposed through the C ABI (tests/test_ops_gpu.py): claims of the form "this rewritten function returns the
// library's bits for EVERY input" are checked over all 2^32 float patterns on the GPU, which takes well under a second.
#include "common.h"

namespace w2v2 {

namespace {

// erf_select == erff and tanh_select == tanhf (common.h), bit for bit; NaN results compare equal whatever their payload.
__global__ __launch_bounds__(256) void select_forms_kernel(unsigned long long* mismatches /* [2]: erf, tanh */) {
    unsigned long long bad_erf = 0, bad_tanh = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((unsigned)i);
        const float a = erff(x), b = erf_select(x);
        if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) ++bad_erf;
        const float c = tanhf(x), d = tanh_select(x);
        if (__float_as_uint(c) != __float_as_uint(d) && !(c != c && d != d)) ++bad_tanh;
    }
    if (bad_erf) atomicAdd(&mismatches[0], bad_erf);
    if (bad_tanh) atomicAdd(&mismatches[1], bad_tanh);
}

}  // namespace

int launch_check_select_forms(unsigned long long* mismatches_dev, hipStream_t s) {
    W2V2_REQUIRE(mismatches_dev, "check_select_forms: null argument");
    W2V2_HIP_CHECK(hipMemsetAsync(mismatches_dev, 0, 2 * sizeof(unsigned long long), s));
    W2V2_LAUNCH(select_forms_kernel, dim3(4096), dim3(256), 0, s, mismatches_dev);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace w2v2
