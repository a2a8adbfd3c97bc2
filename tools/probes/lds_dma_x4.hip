// Where the lanes of a 16-byte LDS-DMA load (global_load_lds_dwordx4, gfx950) put their data: expected M0 + lane * 16.
// build: hipcc -O3 --offload-arch=gfx950 -o lds_dma_x4 lds_dma_x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* g, float* out) {
    __shared__ float s[256];
    s[threadIdx.x] = -1.f; s[threadIdx.x + 64] = -1.f; s[threadIdx.x + 128] = -1.f; s[threadIdx.x + 192] = -1.f;
    __syncthreads();
    if (threadIdx.x < 17)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 2 + threadIdx.x * 4),   // 8-byte aligned only
                                         (__attribute__((address_space(3))) void*)(s + 8), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = s[i];
}
int main() {
    float h[512], *d, *o, r[256];
    for (int i = 0; i < 512; i++) h[i] = (float)i;
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; i++) {
        const float want = (i >= 8 && i < 8 + 68) ? (float)(2 + i - 8) : -1.f;
        if (r[i] != want) { if (bad < 8) printf("s[%d] = %g, expected %g\n", i, r[i], want); bad++; }
    }
    printf("%s (%d mismatches)\n", bad ? "UNEXPECTED" : "lane i of a dwordx4 LDS-DMA load writes M0 + 16 i: ok", bad);
    return bad != 0;
}
