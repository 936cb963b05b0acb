// Microbenchmark 2: start from the pure MFMA + LDS-fragment-read loop (155 TF) and add, one at a time, what the
// real GEMM K loop has: the per-K-tile barrier, the LDS-DMA of the next tile, buffer alternation.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) float lds_f32;
typedef const __attribute__((address_space(1))) float glb_f32;

// FLAGS bit0: barrier per K tile; bit1: DMA next tile (8 KiB A + 8 KiB B per 4 waves... 32 KiB per tile); bit2: alternate buffers
template <int FLAGS, int NWAVE>
__global__ __launch_bounds__(NWAVE * 64, 2) void k(float* out, const float* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // 2 x 8192 floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 16384; i += NWAVE * 64) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        lds[i] = (float)(h & 0xFFFFFF) / 8388608.0f - 1.0f;
    }
    __syncthreads();
    constexpr int MT = NWAVE == 4 ? 2 : 2, NTL = NWAVE == 4 ? 2 : 1;
    const int wm = NWAVE == 4 ? wave >> 1 : wave >> 2, wn = NWAVE == 4 ? wave & 1 : wave & 3;
    f32x16 acc[MT][NTL];
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NTL; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* gsrc = src + (size_t)blockIdx.x * 4096 + lane * 4;
    constexpr int PPW = 16 / NWAVE;
    for (int it = 0; it < iters; ++it) {
        const int buf = (FLAGS & 4) ? (it & 1) : 0;
        if ((FLAGS & 2) && !(FLAGS & 8)) {
#pragma unroll
            for (int i = 0; i < 2 * PPW; ++i)
                __builtin_amdgcn_global_load_lds((glb_f32*)(gsrc + ((it * 7 + i) & 15) * 256), (lds_f32*)(lds + (buf ^ 1) * 8192 + (wave * 2 * PPW + i) * 256), 16, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float* As = lds + buf * 8192;
        const float* Bs = lds + buf * 8192 + 4096 + (4 * lh) * 128 + wn * (NTL * 32) + li;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 a[MT]; float b[NTL][4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int i = wm * (MT * 32) + mt * 32 + li;
                a[mt] = *reinterpret_cast<const f32x4*>(As + i * 32 + (((2 * kb + lh) ^ ((i >> 1) & 7)) << 2));
            }
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) b[nt][e] = Bs[(kb * 8 + e) * 128 + nt * 32];
            if ((FLAGS & 2) && (FLAGS & 8)) {       // spread: 2*PPW/4 pieces per k block, issued next to the MFMAs
#pragma unroll
                for (int i = kb * (2 * PPW / 4); i < (kb + 1) * (2 * PPW / 4); ++i)
                    __builtin_amdgcn_global_load_lds((glb_f32*)(gsrc + ((it * 7 + i) & 15) * 256), (lds_f32*)(lds + (buf ^ 1) * 8192 + (wave * 2 * PPW + i) * 256), 16, 0, 0);
            }
            if (FLAGS & 16) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTL; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][e], b[nt][e], acc[mt][nt], 0, 0, 0);
            if (FLAGS & 16) __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (FLAGS & 1) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NTL; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * NWAVE * 64 + tid] = s;
}

template <int FLAGS, int NWAVE>
void run(const char* name, float* out, float* src) {
    const int blocks = 512, iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<FLAGS, NWAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FLAGS, NWAVE>), dim3(blocks), dim3(NWAVE * 64), 65536, 0, out, src, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<FLAGS, NWAVE>), dim3(blocks), dim3(NWAVE * 64), 65536, 0, out, src, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4.0 * iters * 64 * 4096.0;     // 128x128x32 tile per iteration
    printf("%-52s waves/blk=%d  %8.3f ms  %.1f TFLOP/s\n", name, NWAVE, ms, flops / ms / 1e9);
}
int main() {
    float *out, *src; (void)hipMalloc(&out, 512 * 512 * 4); (void)hipMalloc(&src, (512 * 4096 + 8192) * 4); (void)hipMemset(src, 0, (512 * 4096 + 8192) * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<1, 4>("barrier only", out, src);
        run<7, 4>("barrier + DMA burst + alternation (= real K loop)", out, src);
        run<15, 4>("... DMA spread over k blocks", out, src);
        run<23, 4>("... DMA burst + setprio(1) around MFMAs", out, src);
        run<31, 4>("... DMA spread + setprio", out, src);
        run<7, 8>("barrier + DMA burst + alternation", out, src);
        run<15, 8>("... DMA spread over k blocks", out, src);
        run<31, 8>("... DMA spread + setprio", out, src);
    }
    return 0;
}
