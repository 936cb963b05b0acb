// Microbenchmark 3: K-loop proxy for a BMxBN x32 tile with WM x WN waves (each wave 64x64), LDS-DMA double buffer.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) float lds_f32;
typedef const __attribute__((address_space(1))) float glb_f32;

template <int BM, int BN, int WM, int WN, int DMA>
__global__ __launch_bounds__(WM* WN * 64) void k(float* out, const float* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = WM * WN, STAGE = (BM + BN) * 32, NPIECE = STAGE / 256, PPW = NPIECE / NW;
    constexpr int MT = BM / WM / 32, NTL = BN / WN / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 2 * STAGE; i += NW * 64) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        lds[i] = (float)(h & 0xFFFFFF) / 8388608.0f - 1.0f;
    }
    __syncthreads();
    const int wm = wave / WN, wn = wave % WN;
    f32x16 acc[MT][NTL];
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NTL; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* gsrc = src + (size_t)blockIdx.x * 8192 + lane * 4;
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        if (DMA) {
#pragma unroll
            for (int i = 0; i < PPW; ++i)
                __builtin_amdgcn_global_load_lds((glb_f32*)(gsrc + ((it * 7 + i) & 31) * 256), (lds_f32*)(lds + (buf ^ 1) * STAGE + (wave * PPW + i) * 256), 16, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float* As = lds + buf * STAGE;
        const float* Bs = lds + buf * STAGE + BM * 32 + (4 * lh) * BN + wn * (NTL * 32) + li;
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
                for (int e = 0; e < 4; ++e) b[nt][e] = Bs[(kb * 8 + e) * BN + nt * 32];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTL; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][e], b[nt][e], acc[mt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NTL; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * NW * 64 + tid] = s;
}

template <int BM, int BN, int WM, int WN, int DMA>
void run(const char* name, float* out, float* src, int blocks) {
    const int iters = 2000;
    const int lds = 2 * (BM + BN) * 32 * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<BM, BN, WM, WN, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BM, BN, WM, WN, DMA>), dim3(blocks), dim3(WM * WN * 64), lds, 0, out, src, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<BM, BN, WM, WN, DMA>), dim3(blocks), dim3(WM * WN * 64), lds, 0, out, src, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * iters * 2.0 * BM * BN * 32;
    printf("%-40s %dx%d tile, %2d waves, %d blocks, dma=%d: %8.3f ms  %.1f TFLOP/s\n", name, BM, BN, WM * WN, blocks, DMA, ms, flops / ms / 1e9);
}
int main() {
    float *out, *src; (void)hipMalloc(&out, 1024 * 1024 * 4); (void)hipMalloc(&src, (512 * 8192 + 16384) * 4); (void)hipMemset(src, 0, (512 * 8192 + 16384) * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<128, 128, 2, 2, 1>("128x128 2 blk/CU", out, src, 512);
        run<256, 256, 4, 4, 0>("256x256 1 blk/CU no DMA", out, src, 256);
        run<256, 256, 4, 4, 1>("256x256 1 blk/CU", out, src, 256);
        run<256, 128, 4, 2, 1>("256x128 1 blk/CU (8 waves)", out, src, 256);
        run<256, 128, 4, 2, 0>("256x128 1 blk/CU (8 waves) no DMA", out, src, 256);
        run<128, 256, 2, 4, 1>("128x256 1 blk/CU (8 waves)", out, src, 256);
    }
    return 0;
}
