// membench5.hip -- which population layout streams fastest under the real pull pattern?
// Footprint: 256^3 cells, 19 fp32 populations in, 19 out (2 x 1.27 GB), buffers padded by one
// ghost row / layer on every side so that every shifted read stays inside the allocation.
//   planes : [k][z][y][x]        (current layout)
//   ri     : [z][y][k][x]        (row-interleaved: the 19 runs of one row are contiguous, 19 KB)
//   ti4    : [z][y/4][k][4][x]   (4-row tiles: 4 KB per population, 76 KB per tile)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int W = 256, H = 256, D = 256, HP = H + 2, DP = D + 2;
__constant__ int CX[19] = {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0};
__constant__ int CY[19] = {0, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 1, -1, 1, -1};
__constant__ int CZ[19] = {0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, -1, -1, 1, 1, -1, -1, 1};

template <int SC1> __device__ __forceinline__ void st(float* p, float v) {
    if (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __builtin_nontemporal_store(v, p);
}

// planes, block = one row (256 threads), grid (H, D)
template <int PULL, int SC1> __global__ __launch_bounds__(256) void k_planes(const float* __restrict__ in, float* __restrict__ out, long long plane) {
    const int x = threadIdx.x, y = blockIdx.x + 1, z = blockIdx.y + 1;
    const long long own = ((long long)z * HP + y) * W + x;
    float v[19];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        long long a = own;
        if (PULL) a = ((long long)(z - CZ[k]) * HP + (y - CY[k])) * W + (x - CX[k]);
        v[k] = __builtin_nontemporal_load(in + k * plane + a);
    }
#pragma unroll
    for (int k = 0; k < 19; k++) st<SC1>(out + k * plane + own, v[k] * 1.0001f);
}

// row-interleaved, block = one row
template <int PULL, int SC1> __global__ __launch_bounds__(256) void k_ri(const float* __restrict__ in, float* __restrict__ out, int zrev) {
    const int x = threadIdx.x, y = blockIdx.x + 1, z = (zrev ? D - 1 - (int)blockIdx.y : (int)blockIdx.y) + 1;
    const long long own = ((long long)z * HP + y) * (19 * W) + x;
    float v[19];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        long long a = own + k * W;
        if (PULL) a = ((long long)(z - CZ[k]) * HP + (y - CY[k])) * (19 * W) + k * W + (x - CX[k]);
        v[k] = __builtin_nontemporal_load(in + a);
    }
#pragma unroll
    for (int k = 0; k < 19; k++) st<SC1>(out + own + k * W, v[k] * 1.0001f);
}

// row-interleaved, one wave per row with 4 cells per lane; block = 4 rows
template <int PULL> __global__ __launch_bounds__(256) void k_ri4(const float* __restrict__ in, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = lane * 4, y = blockIdx.x * 4 + wave + 1, z = blockIdx.y + 1;
    const long long own = ((long long)z * HP + y) * (19 * W) + x;
    f4 v[19];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        long long a = own + k * W;
        if (PULL) a = ((long long)(z - CZ[k]) * HP + (y - CY[k])) * (19 * W) + k * W + (x - CX[k]);
        const float* p = in + a;
        if (PULL && CX[k] != 0) {  // dword-aligned 16-byte load
            f4 t; __builtin_memcpy(&t, p, 16); v[k] = t;
        } else v[k] = __builtin_nontemporal_load((const f4*)p);
    }
#pragma unroll
    for (int k = 0; k < 19; k++) __builtin_nontemporal_store(v[k] * 1.0001f, (f4*)(out + own + k * W));
}

// 4-row tiles [tile][k][4][x]: block = one tile, thread t moves the float4 at k*1024 + 4t
__global__ __launch_bounds__(256) void k_ti4(const float* __restrict__ in, float* __restrict__ out) {
    const long long tile = (long long)blockIdx.y * (H / 4) + blockIdx.x;  // 16384 tiles of 19456 floats < allocation
    const long long own = tile * (19 * 4 * W) + threadIdx.x * 4;
    f4 v[19];
#pragma unroll
    for (int k = 0; k < 19; k++) v[k] = __builtin_nontemporal_load((const f4*)(in + own + k * 4 * W));
#pragma unroll
    for (int k = 0; k < 19; k++) __builtin_nontemporal_store(v[k] * 1.0001f, (f4*)(out + own + k * 4 * W));
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
    const long long rows = (long long)DP * HP + 8;         // padded row count (+ slack)
    const long long plane = rows * W + 768;                // planes layout: skewed plane stride
    const long long total = 19 * plane + 4096;             // floats; >= 19 * rows * W for the interleaved layouts
    float *A, *B;
    if (hipMalloc(&A, total * 4) != hipSuccess || hipMalloc(&B, total * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(A, 0, total * 4); (void)hipMemset(B, 0, total * 4);
    const double bytes = 38.0 * W * H * D * 4;
    auto rep = [&](const char* name, float ms) { printf("%-44s %7.1f us %6.3f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12); fflush(stdout); };
    const dim3 grow(H, D), g4(H / 4, D);
    rep("planes copy nt/nt", timeit([&] { k_planes<0, 0><<<grow, 256>>>(A, B, plane); }, 10));
    rep("planes pull nt/nt", timeit([&] { k_planes<1, 0><<<grow, 256>>>(A, B, plane); }, 10));
    rep("planes pull nt/sc1", timeit([&] { k_planes<1, 1><<<grow, 256>>>(A, B, plane); }, 10));
    rep("ri copy nt/nt", timeit([&] { k_ri<0, 0><<<grow, 256>>>(A, B, 0); }, 10));
    rep("ri pull nt/nt", timeit([&] { k_ri<1, 0><<<grow, 256>>>(A, B, 0); }, 10));
    rep("ri pull nt/sc1", timeit([&] { k_ri<1, 1><<<grow, 256>>>(A, B, 0); }, 10));
    {   // A->B then B->A with alternating z direction (what consecutive steps do)
        int flip = 0;
        rep("ri pull nt/nt, A<->B, alternating z", timeit([&] { if (flip) k_ri<1, 0><<<grow, 256>>>(B, A, 1); else k_ri<1, 0><<<grow, 256>>>(A, B, 0); flip ^= 1; }, 10));
    }
    rep("ri4 copy (4 cells/lane)", timeit([&] { k_ri4<0><<<g4, 256>>>(A, B); }, 10));
    rep("ri4 pull (4 cells/lane)", timeit([&] { k_ri4<1><<<g4, 256>>>(A, B); }, 10));
    rep("ti4 copy (4-row tiles, 76 KB contiguous)", timeit([&] { k_ti4<<<g4, 256>>>(A, B); }, 10));
    return 0;
}
