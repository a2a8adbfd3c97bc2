// membench12.hip -- writes are the slow side (19-stream write-only 5.84 TB/s against 7.17 read-only): which
// store shape / cache policy writes the row-interleaved layout fastest?  256^3 cells, 19 floats per cell,
// launches of 64 layers.  W1: one float per lane (the step kernel's shape), W4: four consecutive floats per
// lane (16-byte stores).  Also read-only for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int W = 256, H = 256, D = 256, HP = H + 2;

template <int POLICY> __device__ __forceinline__ void st1(float* p, float v) {
    if (POLICY == 0) __builtin_nontemporal_store(v, p);
    else if (POLICY == 1) *p = v;
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1 write-through
}
template <int POLICY> __global__ __launch_bounds__(256) void k_w1(float* __restrict__ out, int z0, float s) {
    const int x = threadIdx.x, y = blockIdx.x + 1, z = z0 + (int)blockIdx.y + 1;
    float* p = out + ((long long)z * HP + y) * (19 * W) + x;
#pragma unroll
    for (int k = 0; k < 19; k++) st1<POLICY>(p + k * W, s + k);
}
template <int POLICY> __global__ __launch_bounds__(256) void k_w4(float* __restrict__ out, int z0, float s) {
    // 256 threads = 4 rows x 64 lanes, 4 floats per lane
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int y = blockIdx.x * 4 + r + 1, z = z0 + (int)blockIdx.y + 1;
    float* p = out + ((long long)z * HP + y) * (19 * W) + lane * 4;
#pragma unroll
    for (int k = 0; k < 19; k++) {
        f4 v = {s + k, s, s, s};
        if (POLICY == 0) __builtin_nontemporal_store(v, (f4*)(p + k * W));
        else *(f4*)(p + k * W) = v;
    }
}
__global__ __launch_bounds__(256) void k_r1(const float* __restrict__ in, float* __restrict__ sink, int z0) {
    const int x = threadIdx.x, y = blockIdx.x + 1, z = z0 + (int)blockIdx.y + 1;
    const float* p = in + ((long long)z * HP + y) * (19 * W) + x;
    float a = 0;
#pragma unroll
    for (int k = 0; k < 19; k++) a += __builtin_nontemporal_load(p + k * W);
    if (a == 123.456f) sink[0] = a;
}

int main() {
    const long long total = (long long)(D + 2) * HP * 19 * W + 4096;
    float *A, *S;
    if (hipMalloc(&A, total * 4) != hipSuccess || hipMalloc(&S, 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(A, 0, total * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double bytes = 19.0 * W * H * D * 4;
    auto run = [&](const char* name, auto launch) {
        for (int K : {256, 64}) {
            auto pass = [&] { for (int z0 = 0; z0 < D; z0 += K) launch(z0, K); };
            pass(); (void)hipDeviceSynchronize();
            float best = 1e9;
            for (int r = 0; r < 3; r++) {
                (void)hipEventRecord(e0); for (int i = 0; i < 8; i++) pass(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 8; best = ms < best ? ms : best;
            }
            printf("%-34s layers/launch %-3d : %7.1f us %6.3f TB/s\n", name, K, best * 1e3, bytes / (best * 1e-3) / 1e12); fflush(stdout);
        }
    };
    run("read-only, dword nt", [&](int z0, int K) { k_r1<<<dim3(H, K), 256>>>(A, S, z0); });
    run("write-only, dword nt", [&](int z0, int K) { k_w1<0><<<dim3(H, K), 256>>>(A, z0, 1.f); });
    run("write-only, dword plain", [&](int z0, int K) { k_w1<1><<<dim3(H, K), 256>>>(A, z0, 1.f); });
    run("write-only, dword sc1", [&](int z0, int K) { k_w1<2><<<dim3(H, K), 256>>>(A, z0, 1.f); });
    run("write-only, dwordx4 nt", [&](int z0, int K) { k_w4<0><<<dim3(H / 4, K), 256>>>(A, z0, 1.f); });
    run("write-only, dwordx4 plain", [&](int z0, int K) { k_w4<1><<<dim3(H / 4, K), 256>>>(A, z0, 1.f); });
    return 0;
}
