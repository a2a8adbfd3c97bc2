// membench6.hip -- refinements of the row-interleaved layout under the pull pattern (see membench5).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int W = 256, H = 256, D = 256, HP = H + 2, DP = D + 2;
__constant__ int CX[19] = {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0};
__constant__ int CY[19] = {0, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 1, -1, 1, -1};
__constant__ int CZ[19] = {0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, -1, -1, 1, 1, -1, -1, 1};
// slot of population k inside a row chunk when populations are grouped by the row they are pulled from
// order: cz=+1 {5,11,14,15,18}, cz=0 {3,7,10 | 0,1,2 | 4,8,9}, cz=-1 {6,12,13,16,17} (within a group by cy)
__constant__ int POS[19] = {8, 9, 10, 5, 11, 1, 15, 6, 12, 13, 7, 2, 16, 17, 3, 0, 14, 18, 4};

// MODE bit0: x shifts, bit1: y/z shifts, bit2: grouped slots
template <int MODE> __global__ __launch_bounds__(256) void k_ri(const float* __restrict__ in, float* __restrict__ out) {
    const int x = threadIdx.x, y = blockIdx.x + 1, z = blockIdx.y + 1;
    const long long own = ((long long)z * HP + y) * (19 * W) + x;
    float v[19];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        const int slot = (MODE & 4) ? POS[k] : k;
        const int dz = (MODE & 2) ? CZ[k] : 0, dy = (MODE & 2) ? CY[k] : 0, dx = (MODE & 1) ? CX[k] : 0;
        const long long a = ((long long)(z - dz) * HP + (y - dy)) * (19 * W) + slot * W + (x - dx);
        v[k] = __builtin_nontemporal_load(in + a);
    }
#pragma unroll
    for (int k = 0; k < 19; k++) __builtin_nontemporal_store(v[k] * 1.0001f, out + own + ((MODE & 4) ? POS[k] : k) * W);
}
// layer-interleaved [z][k][y][x]
template <int MODE> __global__ __launch_bounds__(256) void k_li(const float* __restrict__ in, float* __restrict__ out) {
    const int x = threadIdx.x, y = blockIdx.x + 1, z = blockIdx.y + 1;
    const long long sub = (long long)HP * W, layer = 19 * sub;
    const long long own = (long long)z * layer + (long long)y * W + x;
    float v[19];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        const int dz = (MODE & 2) ? CZ[k] : 0, dy = (MODE & 2) ? CY[k] : 0, dx = (MODE & 1) ? CX[k] : 0;
        v[k] = __builtin_nontemporal_load(in + (long long)(z - dz) * layer + k * sub + (long long)(y - dy) * W + (x - dx));
    }
#pragma unroll
    for (int k = 0; k < 19; k++) __builtin_nontemporal_store(v[k] * 1.0001f, out + own + k * sub);
}
// row-interleaved, 512-thread block = two rows
template <int MODE> __global__ __launch_bounds__(512) void k_ri2(const float* __restrict__ in, float* __restrict__ out) {
    const int x = threadIdx.x & 255, y = blockIdx.x * 2 + (threadIdx.x >> 8) + 1, z = blockIdx.y + 1;
    const long long own = ((long long)z * HP + y) * (19 * W) + x;
    float v[19];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        const long long a = ((long long)(z - CZ[k]) * HP + (y - CY[k])) * (19 * W) + k * W + (x - CX[k]);
        v[k] = __builtin_nontemporal_load(in + a);
    }
#pragma unroll
    for (int k = 0; k < 19; k++) __builtin_nontemporal_store(v[k] * 1.0001f, out + own + k * W);
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
    const long long total = 19 * rows * W + 4096;  // floats
    float *A, *B;
    if (hipMalloc(&A, total * 4) != hipSuccess || hipMalloc(&B, total * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(A, 0, total * 4); (void)hipMemset(B, 0, total * 4);
    const double bytes = 38.0 * W * H * D * 4;
    auto rep = [&](const char* name, float ms) { printf("%-44s %7.1f us %6.3f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12); fflush(stdout); };
    const dim3 grow(H, D);
    rep("ri copy", timeit([&] { k_ri<0><<<grow, 256>>>(A, B); }, 10));
    rep("ri x shifts only", timeit([&] { k_ri<1><<<grow, 256>>>(A, B); }, 10));
    rep("ri y/z shifts only", timeit([&] { k_ri<2><<<grow, 256>>>(A, B); }, 10));
    rep("ri pull", timeit([&] { k_ri<3><<<grow, 256>>>(A, B); }, 10));
    rep("ri grouped, y/z shifts only", timeit([&] { k_ri<6><<<grow, 256>>>(A, B); }, 10));
    rep("ri grouped pull", timeit([&] { k_ri<7><<<grow, 256>>>(A, B); }, 10));
    rep("ri pull, 2 rows per 512-thread block", timeit([&] { k_ri2<3><<<dim3(H / 2, D), 512>>>(A, B); }, 10));
    rep("li copy", timeit([&] { k_li<0><<<grow, 256>>>(A, B); }, 10));
    rep("li pull", timeit([&] { k_li<3><<<grow, 256>>>(A, B); }, 10));
    return 0;
}
