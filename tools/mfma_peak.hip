// Microbenchmark: what fp32-MFMA rate does this chip sustain (a) from registers only, (b) with the GEMM's
// LDS fragment-read pattern beside the MFMAs.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, int rnd) {
    __shared__ __attribute__((aligned(16))) float lds[128 * 36 + 32 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 128 * 36 + 32 * 128; i += 256) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        lds[i] = rnd ? ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) : (float)(i % 7) * 0.01f;   // full-entropy mantissas vs low-entropy
    }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float a0 = lane * 0.001f, b0 = lane * 0.002f;
    const float* As = lds + ((wave >> 1) * 64 + li) * 36 + 4 * lh;
    const float* Bs = lds + 128 * 36 + (4 * lh) * 128 + (wave & 1) * 64 + li;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 a[2]; float b[2][4];
            if (MODE == 1) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(As + mt * 32 * 36 + kb * 8);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[nt][e] = Bs[(kb * 8 + e) * 128 + nt * 32];
            } else {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) a[mt] = f32x4{a0, a0 + 1, a0 + 2, a0 + 3};
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[nt][e] = b0 + e;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][e], b[nt][e], acc[mt][nt], 0, 0, 0);
        }
        if (MODE == 0) { a0 += 1e-6f; asm volatile("" : "+v"(a0)); }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, int blocks, int rnd, int iters) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, rnd);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, rnd);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 64 * 4096.0;
    printf("%-28s rnd=%d blocks=%4d  %8.3f ms  %.1f TFLOP/s\n", name, rnd, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("registers only, 2 blk/CU", 512, 0, 2000);
        run<1>("LDS fragment reads, 2 blk/CU", 512, 0, 2000);
        run<1>("LDS fragment reads, 2 blk/CU", 512, 1, 2000);
        run<1>("LDS fragment reads, 2 blk/CU", 512, 1, 20000);   // ~70 ms: long enough for DVFS to settle
        run<1>("LDS fragment reads, 2 blk/CU", 512, 0, 20000);
    }
    return 0;
}
