#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr_in, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int a = addr_in[threadIdx.x];    // element index (16-bit units), must be multiple of 4
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)r[j];
}
int main() {
    int h_addr[64]; uint16_t h_out[256];
    int *d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int pat = 0; pat < 2; ++pat) {
        for (int l = 0; l < 64; ++l) h_addr[l] = pat == 0 ? l * 4 : (l % 16) * 64 + (l / 16) * 4;   // pattern 0: lane-linear; 1: lane i of a group -> row i (64 elems apart), group g -> cols 4g
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h_addr[l], h_out[4*l], h_out[4*l+1], h_out[4*l+2], h_out[4*l+3]);
    }
    return 0;
}
