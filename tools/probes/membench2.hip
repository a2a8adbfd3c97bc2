// membench2.hip -- sweeps of the 19-in/19-out streaming pattern (plane skew, A/B offset, read-only /
// write-only, cells per thread).  Tuning probe, not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int Q, int MODE> __global__ void k_dw(const float* __restrict__ in, float* __restrict__ out, long long n, long long plane) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[Q];
    float acc = 0;
#pragma unroll
    for (int k = 0; k < Q; k++) { v[k] = (MODE != 2) ? __builtin_nontemporal_load(in + k * plane + i) : (float)k; acc += v[k]; }
    if (MODE == 1) { if (acc == 12345.678f) out[i] = acc; return; }
#pragma unroll
    for (int k = 0; k < Q; k++) __builtin_nontemporal_store(v[k] * 1.0001f, out + k * plane + i);
}
// each thread handles R cells strided by blockDim (block covers R*256 contiguous cells per plane)
template <int Q, int R> __global__ void k_dw_r(const float* __restrict__ in, float* __restrict__ out, long long n, long long plane) {
    long long base = (long long)blockIdx.x * blockDim.x * R + threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; r++) {
        long long i = base + (long long)r * blockDim.x;
        if (i >= n) return;
        float v[Q];
#pragma unroll
        for (int k = 0; k < Q; k++) v[k] = __builtin_nontemporal_load(in + k * plane + i);
#pragma unroll
        for (int k = 0; k < Q; k++) __builtin_nontemporal_store(v[k] * 1.0001f, out + k * plane + i);
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
    const long long n = 256ll * 256 * 256;
    const long long maxplane = n + 2 * 65536 + (1 << 20);
    char *A, *Bf;
    if (hipMalloc(&A, 19 * maxplane * 4 + (8 << 20)) != hipSuccess || hipMalloc(&Bf, 19 * maxplane * 4 + (8 << 20)) != hipSuccess) return 1;
    (void)hipMemset(A, 0, 19 * maxplane * 4 + (8 << 20)); (void)hipMemset(Bf, 0, 19 * maxplane * 4 + (8 << 20));
    const int B = 256;
    auto tb = [&](float ms, double streams) { return streams * n * 4 / (ms * 1e-3) / 1e12; };
    printf("# skew sweep (bytes between planes beyond (n+2 layers)*4), copy19 dword nt\n");
    const long long skews[] = {0, 128, 256, 512, 1024, 1280, 2048, 2304, 3072, 4096, 4352, 5120, 6144, 8192, 8448, 12288, 16384, 16640, 20480, 32768, 33024, 65536, 131072, 262144 + 4352, 1 << 20};
    for (long long sk : skews) {
        long long plane = n + 2 * 65536 + sk / 4;
        float ms = timeit([&] { k_dw<19, 0><<<(n + B - 1) / B, B>>>((float*)A, (float*)Bf, n, plane); }, 10);
        printf("skew %8lld: %7.1f us %6.3f TB/s\n", sk, ms * 1e3, tb(ms, 38)); fflush(stdout);
    }
    long long plane = n + 2 * 65536 + 1024 / 4;
    printf("# A/B relative offset sweep (skew 1024)\n");
    const long long offs[] = {0, 256, 1024, 4096, 8192, 65536, 1 << 20, (1 << 20) + 4096 * 3, 2 << 20, 4 << 20};
    for (long long o : offs) {
        float ms = timeit([&] { k_dw<19, 0><<<(n + B - 1) / B, B>>>((float*)A, (float*)(Bf + o), n, plane); }, 10);
        printf("boff %8lld: %7.1f us %6.3f TB/s\n", o, ms * 1e3, tb(ms, 38)); fflush(stdout);
    }
    printf("# read-only / write-only (19 streams)\n");
    { float ms = timeit([&] { k_dw<19, 1><<<(n + B - 1) / B, B>>>((float*)A, (float*)Bf, n, plane); }, 10); printf("read-only : %7.1f us %6.3f TB/s\n", ms * 1e3, tb(ms, 19)); }
    { float ms = timeit([&] { k_dw<19, 2><<<(n + B - 1) / B, B>>>((float*)A, (float*)Bf, n, plane); }, 10); printf("write-only: %7.1f us %6.3f TB/s\n", ms * 1e3, tb(ms, 19)); }
    printf("# stream count (copy Q in + Q out, footprint scaled by running on same n)\n");
    { float ms = timeit([&] { k_dw<1, 0><<<(n * 19 + B - 1) / B, B>>>((float*)A, (float*)Bf, n * 19, plane); }, 10); printf("Q=1 x19n : %7.1f us %6.3f TB/s\n", ms * 1e3, tb(ms, 38)); }
    { float ms = timeit([&] { k_dw<4, 0><<<(n + B - 1) / B, B>>>((float*)A, (float*)Bf, n, plane); }, 10); printf("Q=4      : %7.1f us %6.3f TB/s\n", ms * 1e3, tb(ms, 8)); }
    { float ms = timeit([&] { k_dw<8, 0><<<(n + B - 1) / B, B>>>((float*)A, (float*)Bf, n, plane); }, 10); printf("Q=8      : %7.1f us %6.3f TB/s\n", ms * 1e3, tb(ms, 16)); }
    printf("# rows per block (R x 256 contiguous cells per plane per block)\n");
    { float ms = timeit([&] { k_dw_r<19, 2><<<(n / 2 + B - 1) / B, B>>>((float*)A, (float*)Bf, n, plane); }, 10); printf("R=2 : %7.1f us %6.3f TB/s\n", ms * 1e3, tb(ms, 38)); }
    { float ms = timeit([&] { k_dw_r<19, 4><<<(n / 4 + B - 1) / B, B>>>((float*)A, (float*)Bf, n, plane); }, 10); printf("R=4 : %7.1f us %6.3f TB/s\n", ms * 1e3, tb(ms, 38)); }
    { float ms = timeit([&] { k_dw_r<19, 8><<<(n / 8 + B - 1) / B, B>>>((float*)A, (float*)Bf, n, plane); }, 10); printf("R=8 : %7.1f us %6.3f TB/s\n", ms * 1e3, tb(ms, 38)); }
    return 0;
}
