// membench9.hip -- is the pull pattern bound by the NUMBER of vector-memory instructions?
// Row-interleaved layout, 256^3; 1, 2 and 4 cells per lane (dword / dwordx2 / dwordx4), block = 256 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int W = 256, H = 256, D = 256, HP = H + 2, DP = D + 2;
__constant__ int CX[19] = {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0};
__constant__ int CY[19] = {0, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 1, -1, 1, -1};
__constant__ int CZ[19] = {0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, -1, -1, 1, 1, -1, -1, 1};

// N cells per lane; a wave covers 64*N consecutive x; block = 256 threads = 256*N cells = N rows
template <int N, int PULL> __global__ __launch_bounds__(256) void k_ri(const float* __restrict__ in, float* __restrict__ out) {
    typedef float vec __attribute__((ext_vector_type(N)));
    const int cell = threadIdx.x * N;          // 0 .. 256*N-1 over N rows
    const int x = cell % W, y = blockIdx.x * N + cell / W + 1, z = blockIdx.y + 1;
    const long long own = ((long long)z * HP + y) * (19 * W) + x;
    vec v[19];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        long long a = own + k * W;
        if (PULL) a = ((long long)(z - CZ[k]) * HP + (y - CY[k])) * (19 * W) + k * W + (x - CX[k]);
        vec t; __builtin_memcpy(&t, in + a, sizeof(vec));   // 4-byte aligned
        v[k] = t;
    }
#pragma unroll
    for (int k = 0; k < 19; k++) { vec t = v[k] * 1.0001f; __builtin_nontemporal_store(t, (vec*)(out + own + k * W)); }
}
template <> __global__ __launch_bounds__(256) void k_ri<1, 1>(const float* __restrict__ in, float* __restrict__ out) {
    const int x = threadIdx.x, y = blockIdx.x + 1, z = blockIdx.y + 1;
    const long long own = ((long long)z * HP + y) * (19 * W) + x;
    float v[19];
#pragma unroll
    for (int k = 0; k < 19; k++) v[k] = __builtin_nontemporal_load(in + ((long long)(z - CZ[k]) * HP + (y - CY[k])) * (19 * W) + k * W + (x - CX[k]));
#pragma unroll
    for (int k = 0; k < 19; k++) __builtin_nontemporal_store(v[k] * 1.0001f, out + own + k * W);
}
// single linear stream, one dword per thread (how fast is the simplest possible copy at dword width?)
__global__ void k_lin(const float* __restrict__ in, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i) * 1.0001f, out + i);
}
template <int U> __global__ void k_lin_u(const float* __restrict__ in, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x * U + threadIdx.x;
    float v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(in + i + (long long)u * blockDim.x);
#pragma unroll
    for (int u = 0; u < U; u++) __builtin_nontemporal_store(v[u] * 1.0001f, out + i + (long long)u * blockDim.x);
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
int main() {
    const long long rows = (long long)DP * HP + 8;
    const long long total = 19 * rows * W + 4096;
    float *A, *B;
    if (hipMalloc(&A, total * 4) != hipSuccess || hipMalloc(&B, total * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(A, 0, total * 4); (void)hipMemset(B, 0, total * 4);
    const double bytes = 38.0 * W * H * D * 4;
    auto rep = [&](const char* name, float ms) { printf("%-52s %7.1f us %6.3f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12); fflush(stdout); };
    const long long n = 19ll * W * H * D;   // floats moved by the linear copies (< total)
    rep("linear, 1 dword per thread", timeit([&] { k_lin<<<(unsigned)((n + 255) / 256), 256>>>(A, B, n); }, 10));
    rep("linear, 19 dwords per thread (stride 1 KB)", timeit([&] { k_lin_u<19><<<(unsigned)(n / (256 * 19)), 256>>>(A, B, n); }, 10));
    rep("linear, 4 dwords per thread (stride 1 KB)", timeit([&] { k_lin_u<4><<<(unsigned)(n / (256 * 4)), 256>>>(A, B, n); }, 10));
    rep("ri pull, 1 cell per lane (nt dword)", timeit([&] { k_ri<1, 1><<<dim3(H, D), 256>>>(A, B); }, 10));
    rep("ri copy, 2 cells per lane (dwordx2)", timeit([&] { k_ri<2, 0><<<dim3(H / 2, D), 256>>>(A, B); }, 10));
    rep("ri pull, 2 cells per lane (dwordx2)", timeit([&] { k_ri<2, 1><<<dim3(H / 2, D), 256>>>(A, B); }, 10));
    rep("ri copy, 4 cells per lane (dwordx4)", timeit([&] { k_ri<4, 0><<<dim3(H / 4, D), 256>>>(A, B); }, 10));
    rep("ri pull, 4 cells per lane (dwordx4)", timeit([&] { k_ri<4, 1><<<dim3(H / 4, D), 256>>>(A, B); }, 10));
    return 0;
}
