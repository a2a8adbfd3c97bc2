// membench10.hip -- x shifts through LDS: every population is loaded ALIGNED (one 256-byte run per wave),
// the ten x-shifted ones are staged in LDS per 256-cell block and read back one cell to the left / right;
// the two block-edge cells per population come from one extra 10-lane gather load per block.
// Against plain misaligned loads (membench6 "ri pull"), real collision included.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../cfd_amd/csrc/d3q19.h"
using namespace d3q19;
constexpr int W = 256, H = 256, D = 256, HP = H + 2, DP = D + 2;

__device__ __forceinline__ long long row_off(int K, long long ownrow) {
    return ownrow - (long long)CZ[K] * HP * (19 * W) - (long long)CY[K] * (19 * W) + K * W;
}
template <int K> struct G {
    static __device__ __forceinline__ void plain(const float* __restrict__ in, long long ownrow, int x, float* f) {
        G<K - 1>::plain(in, ownrow, x, f);
        const float* p = in + row_off(K, ownrow) + x - CX[K];
        f[K] = CX[K] == 0 ? __builtin_nontemporal_load(p) : *p;
    }
    static __device__ __forceinline__ void aligned(const float* __restrict__ in, long long ownrow, int x, float* f) {
        G<K - 1>::aligned(in, ownrow, x, f);
        f[K] = __builtin_nontemporal_load(in + row_off(K, ownrow) + x);
    }
};
template <> struct G<-1> {
    static __device__ __forceinline__ void plain(const float*, long long, int, float*) {}
    static __device__ __forceinline__ void aligned(const float*, long long, int, float*) {}
};
__constant__ int XK[10] = {1, 7, 9, 11, 13, 2, 8, 10, 12, 14};  // first five: cx = +1 (need x-1), last five: cx = -1 (need x+1)

template <int MODE> __global__ __launch_bounds__(256) void k_step(const float* __restrict__ in, float* __restrict__ out) {
    __shared__ float sh[10][W + 2];
    const int x = threadIdx.x;
    const int y = blockIdx.x + 1, z = blockIdx.y + 1;
    const long long ownrow = ((long long)z * HP + y) * (19 * W);
    float f[19];
    if (MODE == 0) {
        G<18>::plain(in, ownrow, x, f);
    } else {
        G<18>::aligned(in, ownrow, x, f);
        float edge = 0.0f;
        if (x < 10) {  // block-edge cells: x = -1 of the cx = +1 populations, x = W of the cx = -1 ones
            const int K = XK[x];
            edge = in[row_off(K, ownrow) + (x < 5 ? -1 : W)];
        }
        constexpr int KP[5] = {1, 7, 9, 11, 13};
#pragma unroll
        for (int j = 0; j < 5; j++) {
            sh[j][1 + x] = f[KP[j]];
            sh[5 + j][1 + x] = f[KP[j] + 1];
        }
        if (x < 10) sh[x][x < 5 ? 0 : W + 1] = edge;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 5; j++) {
            f[KP[j]] = sh[j][x];          // cell x-1
            f[KP[j] + 1] = sh[5 + j][x + 2];  // cell x+1
        }
    }
    float o[19];
    if (MODE == 2) {   // correctness probe: pass the gathered values through
#pragma unroll
        for (int k = 0; k < 19; k++) o[k] = f[k];
    } else {
        const Moments m = moments_of(f);
        collide_cell(f, m, 1.98f, 1.90f, o);
    }
#pragma unroll
    for (int k = 0; k < 19; k++) __builtin_nontemporal_store(o[k], out + ownrow + k * W + x);
}
__global__ __launch_bounds__(256) void k_ref(const float* __restrict__ in, float* __restrict__ out) {
    const int x = threadIdx.x, y = blockIdx.x + 1, z = blockIdx.y + 1;
    const long long ownrow = ((long long)z * HP + y) * (19 * W);
    float f[19];
    G<18>::plain(in, ownrow, x, f);
#pragma unroll
    for (int k = 0; k < 19; k++) out[ownrow + k * W + x] = f[k];
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
    k_fill<<<nb, 256>>>(A, total, 1);
    (void)hipMemset(B, 0, total * 4); (void)hipMemset(C, 0, total * 4);
    k_ref<<<g1, 256>>>(A, B);
    k_step<2><<<g1, 256>>>(A, C);
    (void)hipDeviceSynchronize();
    {
        const size_t nchk = (size_t)1 << 24;
        float* hb = (float*)malloc(nchk * 4); float* hc = (float*)malloc(nchk * 4);
        const long long start = ((long long)1 * HP + 1) * (19 * W);
        (void)hipMemcpy(hb, B + start, nchk * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hc, C + start, nchk * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < nchk; i++) bad += hb[i] != hc[i];
        printf("LDS shift check: %zu mismatches of %zu\n", bad, nchk); fflush(stdout);
        free(hb); free(hc);
    }
    k_fill<<<nb, 256>>>(A, total, 0);
    k_fill<<<nb, 256>>>(B, total, 0);
    const double bytes = 38.0 * W * H * D * 4;
    auto rep = [&](const char* name, float ms) { printf("%-52s %7.1f us %6.3f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12); fflush(stdout); };
    int flip = 0;
#define RUN(MODE, NAME) flip = 0; rep(NAME, timeit([&] { if (flip) k_step<MODE><<<g1, 256>>>(B, A); else k_step<MODE><<<g1, 256>>>(A, B); flip ^= 1; }, 10));
    RUN(0, "misaligned loads (plain policy) for cx != 0")
    RUN(1, "aligned loads + LDS shift + one 10-lane edge load")
    RUN(0, "misaligned loads (again)")
    RUN(1, "aligned loads + LDS shift (again)")
    return 0;
}
