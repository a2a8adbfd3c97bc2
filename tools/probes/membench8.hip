// membench8.hip -- x-shifted populations by aligned loads + a DPP wave shift + 5 two-lane edge loads,
// against plain misaligned loads (row-interleaved layout, real collision).  See membench7.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../cfd_amd/csrc/d3q19.h"
using namespace d3q19;
constexpr int W = 256, H = 256, D = 256, HP = H + 2, DP = D + 2;

__device__ __forceinline__ long long row_off(int K, long long own) {
    return own - (long long)CZ[K] * HP * (19 * W) - (long long)CY[K] * (19 * W) + K * W;
}
template <int K> struct G {
    static __device__ __forceinline__ void plain(const float* __restrict__ in, long long own, float* f) {
        G<K - 1>::plain(in, own, f);
        f[K] = __builtin_nontemporal_load(in + row_off(K, own) - CX[K]);
    }
    // aligned: every lane loads its own x; the shift happens in registers
    static __device__ __forceinline__ void aligned(const float* __restrict__ in, long long own, float* f) {
        G<K - 1>::aligned(in, own, f);
        f[K] = __builtin_nontemporal_load(in + row_off(K, own));
    }
};
template <> struct G<-1> {
    static __device__ __forceinline__ void plain(const float*, long long, float*) {}
    static __device__ __forceinline__ void aligned(const float*, long long, float*) {}
};
__device__ __forceinline__ float wave_shr1(float v, float edge) {  // lane l <- lane l-1, lane 0 <- edge
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float v, float edge) {  // lane l <- lane l+1, lane 63 <- edge
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

template <int MODE> __global__ __launch_bounds__(256) void k_step(const float* __restrict__ in, float* __restrict__ out) {
    const int x = threadIdx.x, lane = threadIdx.x & 63;
    const int y = blockIdx.x + 1, z = blockIdx.y + 1;
    const long long own = ((long long)z * HP + y) * (19 * W) + x;
    float f[19];
    if (MODE == 0) {
        G<18>::plain(in, own, f);
    } else {
        G<18>::aligned(in, own, f);
        // edge values: lane 0 needs x0-1 of the cx=+1 populations, lane 63 needs x0+64 of the cx=-1 ones
        float e[5];
        constexpr int KP[5] = {1, 7, 9, 11, 13};  // KP[j]+1 is the cx=-1 partner
#pragma unroll
        for (int j = 0; j < 5; j++) {
            e[j] = 0.0f;
            if (lane == 0 || lane == 63) {
                const long long a = lane == 0 ? row_off(KP[j], own) - 1 : row_off(KP[j] + 1, own) + 1;
                e[j] = __builtin_nontemporal_load(in + a);
            }
        }
#pragma unroll
        for (int j = 0; j < 5; j++) {
            f[KP[j]] = wave_shr1(f[KP[j]], e[j]);
            f[KP[j] + 1] = wave_shl1(f[KP[j] + 1], e[j]);
        }
    }
    float o[19];
    const Moments m = moments_of(f);
    collide_cell(f, m, 1.98f, 1.90f, o);
#pragma unroll
    for (int k = 0; k < 19; k++) __builtin_nontemporal_store(o[k], out + own + k * W);
}
// correctness of the shift: out[k] = gathered value (no collision)
template <int MODE> __global__ __launch_bounds__(256) void k_check(const float* __restrict__ in, float* __restrict__ out) {
    const int x = threadIdx.x, lane = threadIdx.x & 63;
    const int y = blockIdx.x + 1, z = blockIdx.y + 1;
    const long long own = ((long long)z * HP + y) * (19 * W) + x;
    float f[19];
    if (MODE == 0) G<18>::plain(in, own, f);
    else {
        G<18>::aligned(in, own, f);
        float e[5];
        constexpr int KP[5] = {1, 7, 9, 11, 13};
#pragma unroll
        for (int j = 0; j < 5; j++) {
            e[j] = 0.0f;
            if (lane == 0 || lane == 63) {
                const long long a = lane == 0 ? row_off(KP[j], own) - 1 : row_off(KP[j] + 1, own) + 1;
                e[j] = in[a];
            }
        }
#pragma unroll
        for (int j = 0; j < 5; j++) {
            f[KP[j]] = wave_shr1(f[KP[j]], e[j]);
            f[KP[j] + 1] = wave_shl1(f[KP[j] + 1], e[j]);
        }
    }
#pragma unroll
    for (int k = 0; k < 19; k++) out[own + k * W] = f[k];
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
__global__ void k_fill(float* a, long long n, int rnd) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = rnd ? (float)((i * 2654435761ull) & 0xffff) : weight((int)((i / W) % 19));
}
int main() {
    const long long rows = (long long)DP * HP + 8;
    const long long total = 19 * rows * W + 4096;
    float *A, *B, *C;
    if (hipMalloc(&A, total * 4) != hipSuccess || hipMalloc(&B, total * 4) != hipSuccess || hipMalloc(&C, total * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const unsigned nb = (unsigned)((total + 255) / 256);
    const dim3 g1(H, D);
    // 1. the register shift must reproduce the misaligned loads exactly
    k_fill<<<nb, 256>>>(A, total, 1);
    (void)hipMemset(B, 0, total * 4); (void)hipMemset(C, 0, total * 4);
    k_check<0><<<g1, 256>>>(A, B);
    k_check<1><<<g1, 256>>>(A, C);
    (void)hipDeviceSynchronize();
    {
        const size_t nchk = (size_t)1 << 24;  // first 16 M floats of the written region
        float* hb = (float*)malloc(nchk * 4); float* hc = (float*)malloc(nchk * 4);
        const long long start = ((long long)1 * HP + 1) * (19 * W);
        (void)hipMemcpy(hb, B + start, nchk * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hc, C + start, nchk * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < nchk; i++) bad += hb[i] != hc[i];
        printf("shift check: %zu mismatches of %zu\n", bad, nchk); fflush(stdout);
        free(hb); free(hc);
    }
    k_fill<<<nb, 256>>>(A, total, 0);
    k_fill<<<nb, 256>>>(B, total, 0);
    const double bytes = 38.0 * W * H * D * 4;
    auto rep = [&](const char* name, float ms) { printf("%-52s %7.1f us %6.3f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12); fflush(stdout); };
    int flip = 0;
#define RUN(MODE, NAME) flip = 0; rep(NAME, timeit([&] { if (flip) k_step<MODE><<<g1, 256>>>(B, A); else k_step<MODE><<<g1, 256>>>(A, B); flip ^= 1; }, 10));
    RUN(0, "misaligned loads for cx != 0")
    RUN(1, "aligned loads + DPP wave shift + 5 edge loads")
    RUN(0, "misaligned loads for cx != 0 (again)")
    RUN(1, "aligned loads + DPP wave shift + 5 edge loads (again)")
    return 0;
}
