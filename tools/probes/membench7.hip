// membench7.hip -- what separates the real step kernel (5.5 TB/s) from the bare pull pattern
// (5.9 TB/s, membench6)?  Row-interleaved layout, interior pull gather + the real collision
// (d3q19.h), with the pieces switched on one at a time.
//   MODE bit0: real moments + collision   bit1: per-tile flag byte (dependent load before the gather)
//        bit2: two rows per wave (38 loads in flight)   bit3: sc1 stores instead of nt
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../cfd_amd/csrc/d3q19.h"
using namespace d3q19;
constexpr int W = 256, H = 256, D = 256, DP = D + 2;

template <int K> struct G {
    static __device__ __forceinline__ void go(const float* __restrict__ in, long long own, float* f, int HP) {
        G<K - 1>::go(in, own, f, HP);
        const long long a = own - (long long)CZ[K] * HP * (19 * W) - (long long)CY[K] * (19 * W) + K * W - CX[K];
        f[K] = __builtin_nontemporal_load(in + a);
    }
};
template <> struct G<-1> { static __device__ __forceinline__ void go(const float*, long long, float*, int) {} };

template <int MODE> __device__ __forceinline__ void st(float* p, float v) {
    if (MODE & 8) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __builtin_nontemporal_store(v, p);
}
template <int MODE> __device__ __forceinline__ void finish(const float* f, float* __restrict__ out, long long own) {
    float o[19];
    if (MODE & 1) {
        const Moments m = moments_of(f);
        collide_cell(f, m, 1.98f, 1.90f, o);
    } else {
#pragma unroll
        for (int k = 0; k < 19; k++) o[k] = f[k] * 1.0001f;
    }
#pragma unroll
    for (int k = 0; k < 19; k++) st<MODE>(out + own + k * W, o[k]);
}

template <int MODE> __global__ __launch_bounds__(256) void k_step(const float* __restrict__ in, float* __restrict__ out,
                                                                  const unsigned char* __restrict__ flags, int HP) {
    const int x = threadIdx.x;
    const int rows = (MODE & 4) ? 2 : 1;
    const int y = blockIdx.x * rows + 1, z = blockIdx.y + 1;
    if (MODE & 2) {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (flags[((long long)(z - 1) * H + (y - 1)) * 4 + wave] != 0) return;  // never taken (all zero)
    }
    const long long own = ((long long)z * HP + y) * (19 * W) + x;
    float f[19], g[19];
    G<18>::go(in, own, f, HP);
    if (MODE & 4) G<18>::go(in, own + 19 * W, g, HP);
    finish<MODE>(f, out, own);
    if (MODE & 4) finish<MODE>(g, out, own + 19 * W);
}

template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(a); for (int i = 0; i < reps; i++) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms / reps < best ? ms / reps : best;
    }
    return best;
}
__global__ void k_fill(float* a, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = weight((int)((i / W) % 19));
}
int main() {
    const long long rows = (long long)DP * 280 + 8;
    const long long total = 19 * rows * W + 4096;
    float *A, *B; unsigned char* F;
    if (hipMalloc(&A, total * 4) != hipSuccess || hipMalloc(&B, total * 4) != hipSuccess || hipMalloc(&F, (size_t)D * H * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    k_fill<<<(unsigned)((total + 255) / 256), 256>>>(A, total);
    k_fill<<<(unsigned)((total + 255) / 256), 256>>>(B, total);
    (void)hipMemset(F, 0, (size_t)D * H * 4);
    const double bytes = 38.0 * W * H * D * 4;
    auto rep = [&](const char* name, float ms) { printf("%-52s %7.1f us %6.3f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12); fflush(stdout); };
    const dim3 g1(H, D), g2(H / 2, D);
    int flip = 0, hp = 258;
#define RUN(MODE, GRID, NAME) flip = 0; rep(NAME, timeit([&] { if (flip) k_step<MODE><<<GRID, 256>>>(B, A, F, hp); else k_step<MODE><<<GRID, 256>>>(A, B, F, hp); flip ^= 1; }, 10));
    char name[64];
    for (int hp : {258, 256, 257, 260, 264, 272, 280}) {   // rows per layer incl. padding (y+1 <= 257 needs hp >= 258 only for addressing hygiene; reads stay inside the allocation for any hp >= 256)
        snprintf(name, sizeof name, "pull + collision + flag, %d rows per layer", hp);
        RUN(3, g1, name)
    }
    return 0;
}
