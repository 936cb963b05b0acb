// What the bf16 / fp16 matrix pipe SUSTAINS on this chip, and what it does with fp16 subnormals.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o /tmp/mfma_power_probe && /tmp/mfma_power_probe
//
// (1) A register-only MFMA loop (no LDS, no memory: 2 waves per SIMD, 8 independent 32x32 accumulators per wave) run for ~60 ms per
//     arm, on operands that are all zero, small integers, or random bits: rate, shader clock (s_memtime / s_memrealtime sampled by
//     every wave) and the fraction of the nominal 2.5 PFLOP/s.  If the random-data arm runs at a lower clock than the zero arm, the
//     pipe is power-limited and "MFMA busy x clock" -- not the kernel structure -- bounds every MFMA-dense kernel.
// (2) v_cvt of values in fp16's subnormal range and an fp16 MFMA on subnormal inputs: are they flushed?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-result"

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int TYPE>      // 0 bf16, 1 fp16
__global__ __launch_bounds__(256, 2) void mfma_loop(const u32x4* __restrict__ src, float* out, unsigned long long* clk, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    u32x4 a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = src[(tid * 6 + i) & 65535];
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = src[(tid * 6 + 4 + i) & 65535];
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned long long c0, w0, c1, w1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(w0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (TYPE == 0)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i >> 1]), __builtin_bit_cast(bf16x8, b[i & 1]), acc[i], 0, 0, 0);
                else
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i >> 1]), __builtin_bit_cast(f16x8, b[i & 1]), acc[i], 0, 0, 0);
            }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(w1));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0) {
        clk[2 * (tid >> 6)] = c1 - c0;
        clk[2 * (tid >> 6) + 1] = w1 - w0;
    }
}

__global__ void f16_subnormal_probe(float* out) {
    // conversions
    const float xs[4] = {0x1p-14f, 0x1p-20f, 0x1.8p-24f, 0x1p-25f};
    if (threadIdx.x < 4) {
        const _Float16 h = (_Float16)xs[threadIdx.x];
        out[threadIdx.x] = (float)h;
    }
    // MFMA: A[row][k] = 2^-20 at k = 0 (else 0), B[k][col] = 1 at k = 0 -> C = 2^-20 if subnormal inputs are honoured, 0 if flushed
    f16x8 a = {}, b = {};
    const int lh = threadIdx.x >> 5;
    if (lh == 0) { a[0] = (_Float16)0x1p-20f; b[0] = (_Float16)1.0f; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[4] = c[0];
    // product of two subnormals-free values landing in the fp32 subnormal range is not of interest here; one normal x subnormal:
    f16x8 a2 = {}, b2 = {};
    if (lh == 0) { a2[0] = (_Float16)0x1p-24f; b2[0] = (_Float16)3.0f; }
    f32x16 c2 = {};
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b2, c2, 0, 0, 0);
    if (threadIdx.x == 0) out[5] = c2[0];
}

template <int TYPE>
void arm(const char* name, const u32x4* src, float* out, unsigned long long* clk, int nblocks) {
    const int iters = 6000;      // 6000 x 32 MFMAs x 32 cycles ~ 6.1 M cycles per wave alone, two waves per SIMD: ~5-8 ms per launch
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(mfma_loop<TYPE>, dim3(nblocks), dim3(256), 0, 0, src, out, clk, iters);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop<TYPE>, dim3(nblocks), dim3(256), 0, 0, src, out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * nblocks * 4);
    hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < nblocks * 4; ++i) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
    const double mhz = cyc / wall * 100.0;
    const double flops = (double)reps * nblocks * 4 * (double)iters * 32 * 2.0 * 32 * 32 * 16;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("%-34s %8.2f ms / launch  %8.1f TFLOP/s = %.3f of 2500  clock %6.0f MHz  -> %.3f of the pipe at that clock\n", name, ms / reps, tf, tf / 2500.0, mhz,
           tf / (2500.0 * mhz / 2400.0));
}

int main() {
    const int nblocks = 512;      // 256 CUs x 2 blocks of 4 waves: two waves per SIMD
    u32x4* src; float* out; unsigned long long* clk;
    hipMalloc(&src, 65536 * sizeof(u32x4)); hipMalloc(&out, 64 * sizeof(float)); hipMalloc(&clk, 2 * nblocks * 4 * sizeof(unsigned long long));
    std::vector<uint32_t> h(65536 * 4);
    // zeros
    for (auto& v : h) v = 0;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    arm<0>("bf16, operands all zero", src, out, clk, nblocks);
    arm<1>("fp16, operands all zero", src, out, clk, nblocks);
    // random finite values: bf16 pairs with exponents near 1.0
    srand(1);
    for (auto& v : h) {
        const uint32_t lo = 0x3f00u | (rand() & 0x80ffu), hi = 0x3f00u | (rand() & 0x80ffu);      // +-[0.5, 1): random sign and mantissa
        v = lo | (hi << 16);
    }
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    arm<0>("bf16, random mantissas in [0.5, 1)", src, out, clk, nblocks);
    for (auto& v : h) {
        const uint32_t lo = 0x3800u | (rand() & 0x83ffu), hi = 0x3800u | (rand() & 0x83ffu);      // fp16 +-[0.5, 1)
        v = lo | (hi << 16);
    }
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    arm<1>("fp16, random mantissas in [0.5, 1)", src, out, clk, nblocks);
    // fully random bits with finite exponents (bf16): exponent field 100..140
    for (auto& v : h) {
        auto r16 = [] { return (uint32_t)(((rand() & 1) << 15) | ((100 + rand() % 40) << 7) | (rand() & 0x7f)); };
        v = r16() | (r16() << 16);
    }
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    arm<0>("bf16, random sign/exponent/mantissa", src, out, clk, nblocks);

    float* o; hipMalloc(&o, 8 * sizeof(float));
    hipLaunchKernelGGL(f16_subnormal_probe, dim3(1), dim3(64), 0, 0, o);
    float ho[8];
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    printf("fp32 -> fp16 -> fp32 of 2^-14, 2^-20, 1.5 x 2^-24, 2^-25: %g %g %g %g  (subnormals kept: 2^-20 = %g, 1.5 x 2^-24 -> 2^-23 = %g)\n", ho[0], ho[1], ho[2], ho[3],
           0x1p-20, 0x1p-23);
    printf("fp16 MFMA, A = 2^-20 (subnormal) x B = 1: C = %g (honoured: %g);  A = 2^-24 x B = 3: C = %g (honoured: %g)\n", ho[4], 0x1p-20, ho[5], 3 * 0x1p-24);
    return 0;
}
