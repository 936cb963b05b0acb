// Microbenchmark: issue rate of the integer multiplies the dropout hash uses, against a plain xor, on one wave per SIMD and on
// four.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate
// Each kernel runs 8 independent dependency chains of one instruction type per lane (inline asm, so nothing is folded).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t c) {
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x[i]) : "v"(c));
                if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (OP == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (OP == 3) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (OP == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                if (OP == 5) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
                if (OP == 6) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
                if (OP == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*reinterpret_cast<uint64_t*>(&x[i & 6])) : "v"((uint64_t)c));
            }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= x[i];
    if (s == 0x12345u) out[0] = s;
}

template <int OP>
void run(const char* name, uint32_t* d, int waves_per_simd) {
    const int iters = 2000, blocks = 256 * waves_per_simd;     // 256 threads = one wave per SIMD of a CU
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 3u);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double insts = (double)iters * 64 * waves_per_simd;                 // per SIMD
    printf("%-14s %d wave(s)/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 64);
    for (int w : {1, 4}) {
        run<0>("v_xor_b32", d, w);
        run<1>("v_mul_lo_u32", d, w);
        run<2>("v_mul_u32_u24", d, w);
        run<3>("v_mul_hi_u32", d, w);
        run<5>("v_mad_u32_u24", d, w);
        run<4>("v_exp_f32", d, w);
        run<6>("v_fma_f32", d, w);
        run<7>("v_pk_fma_f32", d, w);
    }
    return 0;
}
