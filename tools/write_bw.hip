// Write-only HBM bandwidth on this chip with the store forms conv0's apply pass could use (3.2 GB, float4 per lane):
//   hipcc --offload-arch=gfx950 -O3 tools/write_bw.hip -o /tmp/wbw && /tmp/wbw
#include <hip/hip_runtime.h>
#include <stdio.h>
using f4 = __attribute__((ext_vector_type(4))) float;
template <int NT> __global__ __launch_bounds__(256) void fill(f4* p, size_t n4, float v) {
    const f4 x = {v, v + 1.f, v + 2.f, v + 3.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(x, p + i); else p[i] = x;
    }
}
// each block writes a contiguous 128 KiB piece (conv0's pattern: 64 frames x 512 channels), grid = pieces
template <int NT> __global__ __launch_bounds__(256) void fill_chunks(f4* p, size_t n4, float v) {
    const f4 x = {v, v + 1.f, v + 2.f, v + 3.f};
    f4* q = p + (size_t)blockIdx.x * 8192;
    for (int i = threadIdx.x; i < 8192; i += 256) { if (NT) __builtin_nontemporal_store(x, q + i); else q[i] = x; }
}
int main() {
    const size_t bytes = (size_t)3224 << 20, n4 = bytes / 16;
    f4* d; hipMalloc(&d, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-42s %7.3f ms  %6.2f TB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
    };
    for (int g : {2048, 8192, 32768}) {
        printf("grid %d\n", g);
        time("  grid-stride float4 plain", [&] { hipLaunchKernelGGL(fill<0>, dim3(g), dim3(256), 0, 0, d, n4, 1.f); });
        time("  grid-stride float4 nontemporal", [&] { hipLaunchKernelGGL(fill<1>, dim3(g), dim3(256), 0, 0, d, n4, 1.f); });
    }
    time("128 KiB chunk per block, plain", [&] { hipLaunchKernelGGL(fill_chunks<0>, dim3(n4 / 8192), dim3(256), 0, 0, d, n4, 1.f); });
    time("128 KiB chunk per block, nontemporal", [&] { hipLaunchKernelGGL(fill_chunks<1>, dim3(n4 / 8192), dim3(256), 0, 0, d, n4, 1.f); });
    hipMemsetAsync(d, 0, bytes, 0); hipDeviceSynchronize();
    time("hipMemsetAsync", [&] { hipMemsetAsync(d, 0, bytes, 0); });
    return 0;
}
