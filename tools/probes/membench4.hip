// membench4.hip -- does a row-interleaved layout ([row][k][x], one block per row) stream faster than 19 planes?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
// big single-stream copy, 16 B per lane, U float4 per thread
template <int U> __global__ void copy_f4(const f4* __restrict__ in, f4* __restrict__ out, long long n4) {
    long long i = ((long long)blockIdx.x * blockDim.x) * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (i + u * blockDim.x < n4) v[u] = __builtin_nontemporal_load(in + i + u * blockDim.x);
#pragma unroll
    for (int u = 0; u < U; u++) if (i + u * blockDim.x < n4) __builtin_nontemporal_store(v[u] * 1.0001f, out + i + u * blockDim.x);
}
// row-interleaved: block = one row of W=256 cells (256 threads), 19 planes of that row contiguous (19*Wp floats)
template <int SC1> __global__ void copy_rowint(const float* __restrict__ in, float* __restrict__ out, int Wp) {
    const long long base = (long long)blockIdx.x * 19 * Wp + threadIdx.x;
    float v[19];
#pragma unroll
    for (int k = 0; k < 19; k++) v[k] = __builtin_nontemporal_load(in + base + k * Wp);
#pragma unroll
    for (int k = 0; k < 19; k++) {
        if (SC1) __hip_atomic_store(out + base + k * Wp, v[k] * 1.0001f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __builtin_nontemporal_store(v[k] * 1.0001f, out + base + k * Wp);
    }
}
// plane layout reference
template <int SC1> __global__ void copy_planes(const float* __restrict__ in, float* __restrict__ out, long long n, long long plane) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[19];
#pragma unroll
    for (int k = 0; k < 19; k++) v[k] = __builtin_nontemporal_load(in + k * plane + i);
#pragma unroll
    for (int k = 0; k < 19; k++) {
        if (SC1) __hip_atomic_store(out + k * plane + i, v[k] * 1.0001f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __builtin_nontemporal_store(v[k] * 1.0001f, out + k * plane + i);
    }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) { (void)hipEventRecord(a); for (int i = 0; i < reps; i++) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms / reps < best ? ms / reps : best; }
    return best;
}
int main() {
    const long long n = 256ll * 256 * 256, plane = n + 2 * 65536 + 768;
    float *A, *B;
    if (hipMalloc(&A, 19 * plane * 4) != hipSuccess || hipMalloc(&B, 19 * plane * 4) != hipSuccess) return 1;
    (void)hipMemset(A, 0, 19 * plane * 4); (void)hipMemset(B, 0, 19 * plane * 4);
    const double bytes = 38.0 * n * 4;
    auto rep = [&](const char* name, float ms) { printf("%-36s %7.1f us %6.3f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12); fflush(stdout); };
    const long long n4 = 19 * n / 4;
    rep("single stream f4 U=1", timeit([&] { copy_f4<1><<<(n4 + 255) / 256, 256>>>((f4*)A, (f4*)B, n4); }, 10));
    rep("single stream f4 U=4", timeit([&] { copy_f4<4><<<(n4 + 1023) / 1024, 256>>>((f4*)A, (f4*)B, n4); }, 10));
    rep("single stream f4 U=8", timeit([&] { copy_f4<8><<<(n4 + 2047) / 2048, 256>>>((f4*)A, (f4*)B, n4); }, 10));
    rep("planes nt/nt", timeit([&] { copy_planes<0><<<(n + 255) / 256, 256>>>(A, B, n, plane); }, 10));
    rep("planes nt/sc1", timeit([&] { copy_planes<1><<<(n + 255) / 256, 256>>>(A, B, n, plane); }, 10));
    rep("row-interleaved nt/nt Wp=256", timeit([&] { copy_rowint<0><<<65536, 256>>>(A, B, 256); }, 10));
    rep("row-interleaved nt/sc1 Wp=256", timeit([&] { copy_rowint<1><<<65536, 256>>>(A, B, 256); }, 10));
    (void)hipMemcpy(B, A, 19 * n * 4, hipMemcpyDeviceToDevice);
    rep("hipMemcpyDtoD", timeit([&] { (void)hipMemcpyAsync(B, A, 19 * n * 4, hipMemcpyDeviceToDevice, 0); }, 10));
    return 0;
}
