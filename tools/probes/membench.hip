// membench.hip -- ceiling probe for the access pattern of the D3Q19 kernel on MI355X:
// Q read streams + Q write streams, plane-major, various widths per lane.  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template <int Q, int NT> __global__ void copy_dw(const float* __restrict__ in, float* __restrict__ out, long long n, long long plane) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) v[k] = NT ? __builtin_nontemporal_load(in + k * plane + i) : in[k * plane + i];
#pragma unroll
    for (int k = 0; k < Q; k++) { if (NT) __builtin_nontemporal_store(v[k] * 1.0001f, out + k * plane + i); else out[k * plane + i] = v[k] * 1.0001f; }
}
template <int Q, int NT> __global__ void copy_f2(const float* __restrict__ in, float* __restrict__ out, long long n, long long plane) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    f2 v[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) v[k] = *(const f2*)(in + k * plane + i);
#pragma unroll
    for (int k = 0; k < Q; k++) { f2 o = v[k] * 1.0001f; if (NT) __builtin_nontemporal_store(o, (f2*)(out + k * plane + i)); else *(f2*)(out + k * plane + i) = o; }
}
template <int Q, int NT> __global__ void copy_f4(const float* __restrict__ in, float* __restrict__ out, long long n, long long plane) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    f4 v[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) v[k] = *(const f4*)(in + k * plane + i);
#pragma unroll
    for (int k = 0; k < Q; k++) { f4 o = v[k] * 1.0001f; if (NT) __builtin_nontemporal_store(o, (f4*)(out + k * plane + i)); else *(f4*)(out + k * plane + i) = o; }
}
// shifted reads like the pull scheme: plane k read at offset i - sh[k]
template <int Q> __global__ void pull_dw(const float* __restrict__ in, float* __restrict__ out, long long n, long long plane, int W, int WH) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) {
        const int cx = (k % 3) - 1, cy = ((k / 3) % 3) - 1, cz = ((k / 9) % 3) - 1;
        long long j = i - cx - (long long)cy * W - (long long)cz * WH;
        j = j < 0 ? 0 : (j >= n ? n - 1 : j);
        v[k] = in[k * plane + j];
    }
#pragma unroll
    for (int k = 0; k < Q; k++) __builtin_nontemporal_store(v[k] * 1.0001f, out + k * plane + i);
}

template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) { hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); best = ms / reps < best ? ms / reps : best; }
    return best;
}

int main() {
    const int W = 256, H = 256, D = 256;
    const long long n = (long long)W * H * D, plane = n + 2 * W * H + 1088;
    float *a, *b;
    CHK(hipMalloc(&a, 19 * plane * 4)); CHK(hipMalloc(&b, 19 * plane * 4));
    CHK(hipMemset(a, 0, 19 * plane * 4)); CHK(hipMemset(b, 0, 19 * plane * 4));
    auto rep = [&](const char* name, float ms, int q) { printf("%-28s %8.1f us  %7.3f TB/s\n", name, ms * 1e3, q * 2.0 * n * 4 / (ms * 1e-3) / 1e12); fflush(stdout); };
    const int B = 256;
    rep("copy Q=1 f4", timeit([&] { copy_f4<1, 0><<<(n / 4 + B - 1) / B, B>>>(a, b, n, plane); }, 50), 1);
    rep("copy Q=1 f4 nt", timeit([&] { copy_f4<1, 1><<<(n / 4 + B - 1) / B, B>>>(a, b, n, plane); }, 50), 1);
    rep("copy Q=19 dword", timeit([&] { copy_dw<19, 0><<<(n + B - 1) / B, B>>>(a, b, n, plane); }, 20), 19);
    rep("copy Q=19 dword nt", timeit([&] { copy_dw<19, 1><<<(n + B - 1) / B, B>>>(a, b, n, plane); }, 20), 19);
    rep("copy Q=19 f2", timeit([&] { copy_f2<19, 0><<<(n / 2 + B - 1) / B, B>>>(a, b, n, plane); }, 20), 19);
    rep("copy Q=19 f2 nt-store", timeit([&] { copy_f2<19, 1><<<(n / 2 + B - 1) / B, B>>>(a, b, n, plane); }, 20), 19);
    rep("copy Q=19 f4", timeit([&] { copy_f4<19, 0><<<(n / 4 + B - 1) / B, B>>>(a, b, n, plane); }, 20), 19);
    rep("copy Q=19 f4 nt-store", timeit([&] { copy_f4<19, 1><<<(n / 4 + B - 1) / B, B>>>(a, b, n, plane); }, 20), 19);
    rep("pull Q=19 dword nt-store", timeit([&] { pull_dw<19><<<(n + B - 1) / B, B>>>(a, b, n, plane, W, W * H); }, 20), 19);
    rep("copy Q=19 f4 B=64", timeit([&] { copy_f4<19, 1><<<(n / 4 + 63) / 64, 64>>>(a, b, n, plane); }, 20), 19);
    rep("copy Q=19 f4 B=1024", timeit([&] { copy_f4<19, 1><<<(n / 4 + 1023) / 1024, 1024>>>(a, b, n, plane); }, 20), 19);
    rep("copy Q=19 dword B=1024", timeit([&] { copy_dw<19, 1><<<(n + 1023) / 1024, 1024>>>(a, b, n, plane); }, 20), 19);
    return 0;
}
