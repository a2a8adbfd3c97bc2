// membench3.hip -- store/load cache-policy variants and in-place update for the 19+19 stream pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LP> __device__ __forceinline__ float ld(const float* p) {
    float v;
    if (LP == 0) v = *p;
    else if (LP == 1) v = __builtin_nontemporal_load(p);
    else if (LP == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int SP> __device__ __forceinline__ void st(float* p, float v) {
    if (SP == 0) *p = v;
    else if (SP == 1) __builtin_nontemporal_store(v, p);
    else if (SP == 2) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (SP == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (SP == 4) asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}
template <int Q, int LP, int SP> __global__ void k(const float* in, float* out, long long n, long long plane) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) v[q] = ld<LP>(in + q * plane + i);
#pragma unroll
    for (int q = 0; q < Q; q++) st<SP>(out + q * plane + i, v[q] * 1.0001f);
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
    const int T = 256; const unsigned G = (n + T - 1) / T;
#define RUN(LP, SP, OUT, name) { float ms = timeit([&] { k<19, LP, SP><<<G, T>>>(A, OUT, n, plane); }, 10); printf("%-40s %7.1f us %6.3f TB/s\n", name, ms * 1e3, 38.0 * n * 4 / (ms * 1e-3) / 1e12); fflush(stdout); }
    RUN(0, 0, B, "A->B ld plain  st plain");
    RUN(0, 1, B, "A->B ld plain  st nt");
    RUN(1, 1, B, "A->B ld nt     st nt");
    RUN(1, 0, B, "A->B ld nt     st plain");
    RUN(1, 2, B, "A->B ld nt     st sc1");
    RUN(1, 3, B, "A->B ld nt     st sc0 sc1");
    RUN(1, 4, B, "A->B ld nt     st sc0 sc1 nt");
    RUN(1, 5, B, "A->B ld nt     st sc0");
    RUN(0, 0, A, "in-place ld plain st plain");
    RUN(0, 1, A, "in-place ld plain st nt");
    RUN(1, 1, A, "in-place ld nt    st nt");
    RUN(1, 3, A, "in-place ld nt    st sc0 sc1");
    return 0;
}
