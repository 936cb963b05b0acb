// How long does the chip take to START and RETIRE N workgroups that do (almost) nothing, with the LDS / thread footprint of
// the bf16 GEMM (64 KiB dynamic LDS, 256 threads, 2 per CU)?  hipcc --offload-arch=gfx950 -O3 tools/empty_dispatch.hip -o /tmp/ed
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256, 2) void k(float* out, int spin) {
    extern __shared__ float s[];
    s[threadIdx.x] = (float)spin;
    __syncthreads();
    float v = s[(threadIdx.x + 1) & 255];
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    if (v == 123.456f) out[0] = v;
}
int main() {
    float* d; hipMalloc(&d, 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int lds : {1024, 65536})
        for (int n : {512, 4096, 24704, 175000})
            for (int spin : {0, 2000}) {
                hipLaunchKernelGGL(k, dim3(n), dim3(256), lds, 0, d, spin);
                hipDeviceSynchronize();
                hipEventRecord(a);
                for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(n), dim3(256), lds, 0, d, spin);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                printf("lds %6d blocks %7d spin %5d : %8.1f us per launch = %6.1f ns per block\n", lds, n, spin, ms * 200.f, ms * 2e5f / n);
            }
    return 0;
}
