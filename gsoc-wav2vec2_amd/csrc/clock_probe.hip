// Shader-clock probe: what frequency does the chip hold while a given kernel mix runs?
//
// MI355X clocks to its power budget (MI355X_MICROARCH.md, "DVFS give-back"): under the fp32 MFMA GEMM it does not hold the
// 2.4 GHz the peak figures are quoted at.  The roofline fraction in bench.py keeps the nominal peak; this probe lets the same
// line also state the clock the kernels actually ran at.  One wave samples s_memtime (one tick per shader cycle) and
// s_memrealtime (constant-rate wall clock, hipDeviceAttributeWallClockRate) at its start, spins for `spin_us` of wall time and
// samples both again: shader MHz = d(memtime) / d(memrealtime) x wall-clock rate.  Launched on a side stream while the workload
// runs on the main one, it occupies one wave slot of one CU.
#include "common.h"

namespace w2v2 {
namespace {

__global__ __launch_bounds__(64) void clock_probe_kernel(uint64_t* out, uint64_t wall_ticks, uint64_t max_iters) {
    if (threadIdx.x != 0) return;
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    uint64_t r1 = r0, it = 0;
    while (r1 - r0 < wall_ticks && it < max_iters) {       // bounded twice: by wall time and by an iteration cap
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
        ++it;
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    out[0] = c0; out[1] = r0; out[2] = c1; out[3] = r1;
}

}  // namespace
}  // namespace w2v2

using namespace w2v2;

extern "C" int w2v2_clock_probe(void* stream, int32_t spin_us, uint64_t* dev_out4, int32_t* wall_clock_khz) {
    W2V2_REQUIRE(dev_out4 && wall_clock_khz, "clock_probe: null argument");
    W2V2_REQUIRE(spin_us > 0 && spin_us <= 200000, "clock_probe: spin_us %d outside (0, 200000]", spin_us);
    int dev = 0, khz = 0;
    W2V2_HIP_CHECK(hipGetDevice(&dev));
    W2V2_HIP_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    W2V2_REQUIRE(khz > 0, "clock_probe: the device reports no wall-clock rate");
    *wall_clock_khz = khz;
    const uint64_t ticks = (uint64_t)spin_us * (uint64_t)khz / 1000u;
    // s_sleep 32 = 2048 cycles per iteration at most: 200 ms of spinning at 2.4 GHz is < 1e6 iterations; cap at 8e6
    W2V2_LAUNCH(clock_probe_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), dev_out4, ticks, (uint64_t)8000000);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}
