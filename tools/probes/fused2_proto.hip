// fused2_proto.hip -- PROTOTYPE, not product: two time steps per pass over HBM (in-kernel temporal
// blocking), interior cells only, to measure what such a kernel can reach on MI355X before the
// boundary chain, slabs and the rest are worth building around it.
//
// A workgroup of 15 waves owns a 64 x 12 tile of (x, y) and marches in z.  Waves 0..13 hold one row
// each of the tile plus one halo row below and above; wave 14 holds the two halo columns
// (x0-1 and x0+64) of those 14 rows.  Per layer L every wave computes the FIRST step for its cells
// (19 pulled loads from the time-t buffer, collision) and keeps the result in registers.  The values
// its neighbours need for the SECOND step of layer z = L-1 -- the 4 up-going oblique populations of
// layer z-1, the 8 in-plane ones of layer z, the 4 down-going oblique ones of layer z+1 -- go through
// LDS (16 floats per haloed cell, 59 KB), the own-column ones (rest, +z, -z) stay in registers.  The
// 12 inner row waves then pull the second step from LDS, collide and store time t+2.
// Traffic per two updates: 19 loads (x 15/12 halo) + 19 stores instead of 38 + 38.
//
// Checked bit for bit against two passes of a plain one-step kernel on every cell further than two
// cells from the domain faces (the prototype has no boundary conditions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../cfd_amd/csrc/d3q19.h"
using namespace d3q19;

constexpr int W = 256, H = 256, D = 256;
constexpr int WP = W + 64, XOFF = 32;                 // x = -2 .. W+1 stay inside a row
constexpr long long ROW = 19ll * WP, LAYER = (long long)H * ROW, TOTAL = (long long)D * LAYER;
#ifndef TILE_Y
#define TILE_Y 12
#endif
#ifndef PREFETCH
#define PREFETCH 0
#endif
constexpr int TY = TILE_Y, ROWS = TY + 2, NT_Y = (H - 4) / TY;  // tiles start at y = 2
constexpr int NTHREADS = (ROWS + 1) * 64;
constexpr int ZS = 2, ZE = D - 2;                      // layers [ZS, ZE) get the two steps
constexpr float OMEGA_S = 1.98f, OMEGA_B = 1.90f;

__device__ __forceinline__ long long at(int k, int x, int y, int z) {
    return (long long)z * LAYER + (long long)y * ROW + (long long)k * WP + x + XOFF;
}

// plain one-step pull kernel (reference for the check and for the timing baseline)
__global__ __launch_bounds__(256) void k_step1(const float* __restrict__ in, float* __restrict__ out) {
    const int x = threadIdx.x, y = blockIdx.x + 1, z = blockIdx.y + 1;
    float f[Q], o[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) f[k] = __builtin_nontemporal_load(in + at(k, x - CX[k], y - CY[k], z - CZ[k]));
    const Moments m = moments_of(f);
    collide_cell(f, m, OMEGA_S, OMEGA_B, o);
#pragma unroll
    for (int k = 0; k < Q; k++) __builtin_nontemporal_store(o[k], out + at(k, x, y, z));
}

// LDS slot of an exchanged population: planes 0..3 up-going oblique (11,14,15,18), 4..11 in-plane
// (1,2,3,4,7,8,9,10), 12..15 down-going oblique (12,13,16,17)
__host__ __device__ constexpr int lds_plane(int k) {
    return k == 11 ? 0 : k == 14 ? 1 : k == 15 ? 2 : k == 18 ? 3 :
           k == 1 ? 4 : k == 2 ? 5 : k == 3 ? 6 : k == 4 ? 7 : k == 7 ? 8 : k == 8 ? 9 : k == 9 ? 10 : k == 10 ? 11 :
           k == 12 ? 12 : k == 13 ? 13 : k == 16 ? 14 : k == 17 ? 15 : -1;
}
constexpr int LCOLS = 66;
constexpr int LDS_FLOATS = 16 * ROWS * LCOLS;

template <int K> struct Put {
    static __device__ __forceinline__ void run(float* lds, int slot, const float* up_prev, const float* cur, const float* nxt) {
        Put<K - 1>::run(lds, slot, up_prev, cur, nxt);
        constexpr int p = lds_plane(K);
        if (p >= 0) lds[p * (ROWS * LCOLS) + slot] = CZ[K] == 1 ? up_prev[K] : (CZ[K] == 0 ? cur[K] : nxt[K]);
    }
};
template <> struct Put<-1> { static __device__ __forceinline__ void run(float*, int, const float*, const float*, const float*) {} };

template <int K> struct Get {
    static __device__ __forceinline__ void run(const float* lds, int r, int col, const float* up_prev, const float* cur, const float* nxt, float* f) {
        Get<K - 1>::run(lds, r, col, up_prev, cur, nxt, f);
        constexpr int p = lds_plane(K);
        if (p >= 0) f[K] = lds[p * (ROWS * LCOLS) + (r - CY[K]) * LCOLS + (col - CX[K])];
        else f[K] = K == 0 ? cur[0] : (K == 5 ? up_prev[5] : nxt[6]);  // own column: registers
    }
};
template <> struct Get<-1> { static __device__ __forceinline__ void run(const float*, int, int, const float*, const float*, const float*, float*) {} };

__global__ __launch_bounds__(NTHREADS) void k_fused2(const float* __restrict__ in, float* __restrict__ out, int zchunk) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * 64, y0 = 2 + blockIdx.y * TY;
    const int zs = ZS + blockIdx.z * zchunk, ze = min(zs + zchunk, ZE);
    // the cell this lane carries through the first step, and its LDS slot
    int x, r;
    bool active = true;
    if (wave < ROWS) { r = wave; x = x0 + lane; }
    else { r = lane % ROWS; x = lane < ROWS ? x0 - 1 : x0 + 64; active = lane < 2 * ROWS; }
    const int y = y0 - 1 + r;
    const int col = x - x0 + 1;
    const int slot = r * LCOLS + col;
    const bool second = wave >= 1 && wave <= TY;  // the 12 inner row waves do the second step

    float up_prev[Q], cur[Q], nxt[Q], pf[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) { up_prev[k] = 0.f; cur[k] = 0.f; nxt[k] = 0.f; pf[k] = 0.f; }
    // software pipeline: the loads of layer L+1 are in flight while layer L-1 gets its second step
    if (PREFETCH && active) {
#pragma unroll
        for (int k = 0; k < Q; k++) pf[k] = __builtin_nontemporal_load(in + at(k, x - CX[k], y - CY[k], zs - 1 - CZ[k]));
    }
    int buf = 0;
    for (int L = zs - 1; L <= ze; L++) {
        if (active) {
            if (!PREFETCH) {
#pragma unroll
                for (int k = 0; k < Q; k++) pf[k] = __builtin_nontemporal_load(in + at(k, x - CX[k], y - CY[k], L - CZ[k]));
            }
            const Moments m = moments_of(pf);
            collide_cell(pf, m, OMEGA_S, OMEGA_B, nxt);
            if (PREFETCH && L < ze) {
#pragma unroll
                for (int k = 0; k < Q; k++) pf[k] = __builtin_nontemporal_load(in + at(k, x - CX[k], y - CY[k], L + 1 - CZ[k]));
            }
        }
        if (L >= zs + 1) {
            const int z = L - 1;
            float* my = lds + buf * LDS_FLOATS;  // two LDS buffers: one barrier per layer
            if (active) Put<Q - 1>::run(my, slot, up_prev, cur, nxt);
            __syncthreads();
            if (second) {
                float f[Q], o[Q];
                Get<Q - 1>::run(my, r, col, up_prev, cur, nxt, f);
                const Moments m = moments_of(f);
                collide_cell(f, m, OMEGA_S, OMEGA_B, o);
#pragma unroll
                for (int k = 0; k < Q; k++) __builtin_nontemporal_store(o[k], out + at(k, x, y, z));
            }
            buf ^= 1;
        }
#pragma unroll
        for (int k = 0; k < Q; k++) { up_prev[k] = cur[k]; cur[k] = nxt[k]; }
    }
}

// Variant B: 8 waves per workgroup, every row wave carries TWO rows (r and r + 7), so a wave may use
// 256 VGPRs: room for the software prefetch of the next layer's 2 x 19 loads, and two independent
// cells per lane to overlap.  Same tile (64 x 12 + halo), same LDS exchange.
constexpr int RB = 14, HALF = 7;
__global__ __launch_bounds__(512) void k_fused2b(const float* __restrict__ in, float* __restrict__ out, int zchunk) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * 64, y0 = 2 + blockIdx.y * 12;
    const int zs = ZS + blockIdx.z * zchunk, ze = min(zs + zchunk, ZE);
    int x, r[2];
    bool act[2], sec[2];
    if (wave < HALF) {
        x = x0 + lane; r[0] = wave; r[1] = wave + HALF; act[0] = act[1] = true;
    } else {  // halo columns: lanes 0..13 -> x0-1, lanes 14..27 -> x0+64, rows 0..13, one cell per lane
        x = lane < RB ? x0 - 1 : x0 + 64; r[0] = lane % RB; r[1] = 0; act[0] = lane < 2 * RB; act[1] = false;
    }
    int y[2], slot[2];
    const int col = x - x0 + 1;
    for (int c = 0; c < 2; c++) {
        y[c] = y0 - 1 + r[c];
        slot[c] = r[c] * LCOLS + col;
        sec[c] = wave < HALF && r[c] >= 1 && r[c] <= 12;
    }
    float up_prev[2][Q], cur[2][Q], nxt[2][Q], pf[2][Q];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int k = 0; k < Q; k++) { up_prev[c][k] = 0.f; cur[c][k] = 0.f; nxt[c][k] = 0.f; pf[c][k] = 0.f; }
#pragma unroll
    for (int c = 0; c < 2; c++)
        if (act[c]) {
#pragma unroll
            for (int k = 0; k < Q; k++) pf[c][k] = __builtin_nontemporal_load(in + at(k, x - CX[k], y[c] - CY[k], zs - 1 - CZ[k]));
        }
    int buf = 0;
    for (int L = zs - 1; L <= ze; L++) {
#pragma unroll
        for (int c = 0; c < 2; c++)
            if (act[c]) {
                const Moments m = moments_of(pf[c]);
                collide_cell(pf[c], m, OMEGA_S, OMEGA_B, nxt[c]);
                if (L < ze) {
#pragma unroll
                    for (int k = 0; k < Q; k++) pf[c][k] = __builtin_nontemporal_load(in + at(k, x - CX[k], y[c] - CY[k], L + 1 - CZ[k]));
                }
            }
        if (L >= zs + 1) {
            const int z = L - 1;
            float* my = lds + buf * (16 * RB * LCOLS);
#pragma unroll
            for (int c = 0; c < 2; c++)
                if (act[c]) Put<Q - 1>::run(my, slot[c], up_prev[c], cur[c], nxt[c]);
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 2; c++)
                if (sec[c]) {
                    float f[Q], o[Q];
                    Get<Q - 1>::run(my, r[c], col, up_prev[c], cur[c], nxt[c], f);
                    const Moments m = moments_of(f);
                    collide_cell(f, m, OMEGA_S, OMEGA_B, o);
#pragma unroll
                    for (int k = 0; k < Q; k++) __builtin_nontemporal_store(o[k], out + at(k, x, y[c], z));
                }
            buf ^= 1;
        }
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int k = 0; k < Q; k++) { up_prev[c][k] = cur[c][k]; cur[c][k] = nxt[c][k]; }
    }
}

// Variant C: as the first kernel (15 waves, one cell per lane), but the 19 loads of the NEXT layer go
// straight from global memory into a wave-private LDS staging area (global_load_lds_dword: no VGPRs
// held while they are in flight), issued as soon as the first step of the current layer has consumed
// the area.  They travel during the exchange, the second step and its stores.
constexpr int STAGE_FLOATS = 15 * 19 * 64;
__global__ __launch_bounds__(960) void k_fused2c(const float* __restrict__ in, float* __restrict__ out, int zchunk) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * 64, y0 = 2 + blockIdx.y * 12;
    const int zs = ZS + blockIdx.z * zchunk, ze = min(zs + zchunk, ZE);
    int x, r;
    bool active = true;
    if (wave < 14) { r = wave; x = x0 + lane; }
    else { r = lane % 14; x = lane < 14 ? x0 - 1 : x0 + 64; active = lane < 28; }
    const int y = y0 - 1 + r;
    const int col = x - x0 + 1;
    const int slot = r * LCOLS + col;
    const bool second = wave >= 1 && wave <= 12;
    float* exch = lds;
    float* stage = lds + 16 * 14 * LCOLS + wave * (19 * 64);  // wave-uniform

    auto issue = [&](int L) {
        if (active) {
#pragma unroll
            for (int k = 0; k < Q; k++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + at(k, x - CX[k], y - CY[k], L - CZ[k])),
                                                 (__attribute__((address_space(3))) void*)(stage + k * 64), 4, 0, 0);
        }
    };
    float up_prev[Q], cur[Q], nxt[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) { up_prev[k] = 0.f; cur[k] = 0.f; nxt[k] = 0.f; }
    issue(zs - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int L = zs - 1; L <= ze; L++) {
        if (active) {
            float f[Q];
#pragma unroll
            for (int k = 0; k < Q; k++) f[k] = stage[k * 64 + lane];
            const Moments m = moments_of(f);
            collide_cell(f, m, OMEGA_S, OMEGA_B, nxt);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging area has been read
        if (L < ze) issue(L + 1);
        if (L >= zs + 1) {
            const int z = L - 1;
            if (active) Put<Q - 1>::run(exch, slot, up_prev, cur, nxt);
            __syncthreads();
            if (second) {
                float f[Q], o[Q];
                Get<Q - 1>::run(exch, r, col, up_prev, cur, nxt, f);
                const Moments m = moments_of(f);
                collide_cell(f, m, OMEGA_S, OMEGA_B, o);
                // the staged loads of the next layer have had the whole second step to land; waiting for
                // them HERE keeps the latency of the stores below out of the next iteration's wait
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < Q; k++) __builtin_nontemporal_store(o[k], out + at(k, x, y, z));
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int k = 0; k < Q; k++) { up_prev[k] = cur[k]; cur[k] = nxt[k]; }
    }
}

static unsigned hash32(unsigned v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; }

int main(int argc, char** argv) {
    const int zsplit = argc > 1 ? atoi(argv[1]) : 3;
    const int zchunk = (ZE - ZS + zsplit - 1) / zsplit;
    float *A, *B, *T, *R;
    const size_t bytes = (size_t)TOTAL * sizeof(float);
    if (hipMalloc(&A, bytes) != hipSuccess || hipMalloc(&B, bytes) != hipSuccess || hipMalloc(&T, bytes) != hipSuccess ||
        hipMalloc(&R, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    {
        std::vector<float> h((size_t)TOTAL);
        for (long long i = 0; i < TOTAL; i++) {
            const int k = (int)((i / WP) % 19);
            h[(size_t)i] = weight(k) * (1.0f + 0.02f * ((hash32((unsigned)i) & 0xffff) / 65535.0f - 0.5f));
        }
        (void)hipMemcpy(A, h.data(), bytes, hipMemcpyHostToDevice);
    }
    (void)hipMemset(B, 0, bytes); (void)hipMemset(T, 0, bytes); (void)hipMemset(R, 0, bytes);
    // every access of both kernels stays inside [0, TOTAL): x in [-2, W+1] with XOFF = 32 and WP = W + 64,
    // y in [0, H), z in [0, D) -- checked here rather than assumed
    if (!(2 + NT_Y * TY + 1 < H && ZE + 1 < D && ZS - 2 >= 0 && XOFF >= 2 && WP >= W + XOFF + 2)) { printf("geometry check failed\n"); return 1; }
    const dim3 gref(H - 2, D - 2), gf(W / 64, NT_Y, zsplit);
    const size_t lds_bytes = 2 * LDS_FLOATS * sizeof(float);
    (void)hipFuncSetAttribute((const void*)k_fused2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);

    k_step1<<<gref, 256>>>(A, T);
    k_step1<<<gref, 256>>>(T, R);
    k_fused2<<<gf, NTHREADS, lds_bytes>>>(A, B, zchunk);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    {
        std::vector<float> hb((size_t)TOTAL), hr((size_t)TOTAL);
        (void)hipMemcpy(hb.data(), B, bytes, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hr.data(), R, bytes, hipMemcpyDeviceToHost);
        long long bad = 0, n = 0;
        for (int z = ZS; z < ZE; z++)
            for (int y = 2; y < 2 + NT_Y * TY; y++)
                for (int k = 0; k < 19; k++)
                    for (int x = 2; x < W - 2; x++) {
                        const size_t i = (size_t)z * LAYER + (size_t)y * ROW + (size_t)k * WP + x + XOFF;
                        n++;
                        if (memcmp(&hb[i], &hr[i], 4) != 0 && bad++ < 5) printf("mismatch k=%d x=%d y=%d z=%d: %g vs %g\n", k, x, y, z, hb[i], hr[i]);
                    }
        printf("check: %lld of %lld interior values differ from two plain steps\n", bad, n);
        if (bad) return 1;
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto timeit = [&](auto f, int reps) {
        f(); (void)hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 3; r++) {
            (void)hipEventRecord(e0); for (int i = 0; i < reps; i++) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms / reps < best ? ms / reps : best;
        }
        return best;
    };
    const double cells_ref = (double)(W) * (H - 2) * (D - 2), cells_f = (double)W * NT_Y * TY * (ZE - ZS);
    const float t1 = timeit([&] { k_step1<<<gref, 256>>>(A, T); k_step1<<<gref, 256>>>(T, R); }, 10);
    printf("two plain steps : %8.1f us  %7.0f MLUPS\n", t1 * 1e3, 2 * cells_ref / (t1 * 1e-3) / 1e6);
    const float t2 = timeit([&] { k_fused2<<<gf, NTHREADS, lds_bytes>>>(A, B, zchunk); }, 10);
    printf("fused two steps, %d rows per tile, prefetch %d : %8.1f us  %7.0f MLUPS  (z split %d, %d workgroups)\n", TY, PREFETCH, t2 * 1e3, 2 * cells_f / (t2 * 1e-3) / 1e6, zsplit,
           gf.x * gf.y * gf.z);
#if TILE_Y == 12
    {
        const size_t lds_c = (size_t)(16 * 14 * LCOLS + STAGE_FLOATS) * sizeof(float);
        (void)hipFuncSetAttribute((const void*)k_fused2c, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
        (void)hipMemset(B, 0, bytes);
        k_fused2c<<<gf, 960, lds_c>>>(A, B, zchunk);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel C failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        std::vector<float> hb((size_t)TOTAL), hr((size_t)TOTAL);
        (void)hipMemcpy(hb.data(), B, bytes, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hr.data(), R, bytes, hipMemcpyDeviceToHost);
        long long bad = 0;
        for (int z = ZS; z < ZE; z++)
            for (int y = 2; y < 2 + NT_Y * TY; y++)
                for (int k = 0; k < 19; k++)
                    for (int x = 2; x < W - 2; x++) {
                        const size_t i = (size_t)z * LAYER + (size_t)y * ROW + (size_t)k * WP + x + XOFF;
                        if (memcmp(&hb[i], &hr[i], 4) != 0) bad++;
                    }
        printf("check C: %lld interior values differ\n", bad);
        if (bad) return 1;
        const float t4 = timeit([&] { k_fused2c<<<gf, 960, lds_c>>>(A, B, zchunk); }, 10);
        printf("fused two steps, variant C (next layer staged in LDS by global_load_lds) : %8.1f us  %7.0f MLUPS\n", t4 * 1e3, 2 * cells_f / (t4 * 1e-3) / 1e6);
    }
    {
        (void)hipFuncSetAttribute((const void*)k_fused2b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipMemset(B, 0, bytes);
        k_fused2b<<<gf, 512, lds_bytes>>>(A, B, zchunk);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel B failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        std::vector<float> hb((size_t)TOTAL), hr((size_t)TOTAL);
        (void)hipMemcpy(hb.data(), B, bytes, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hr.data(), R, bytes, hipMemcpyDeviceToHost);
        long long bad = 0;
        for (int z = ZS; z < ZE; z++)
            for (int y = 2; y < 2 + NT_Y * TY; y++)
                for (int k = 0; k < 19; k++)
                    for (int x = 2; x < W - 2; x++) {
                        const size_t i = (size_t)z * LAYER + (size_t)y * ROW + (size_t)k * WP + x + XOFF;
                        if (memcmp(&hb[i], &hr[i], 4) != 0) bad++;
                    }
        printf("check B: %lld interior values differ\n", bad);
        if (bad) return 1;
        const float t3 = timeit([&] { k_fused2b<<<gf, 512, lds_bytes>>>(A, B, zchunk); }, 10);
        printf("fused two steps, variant B (8 waves, two rows per wave, prefetch) : %8.1f us  %7.0f MLUPS\n", t3 * 1e3, 2 * cells_f / (t3 * 1e-3) / 1e6);
    }
#endif
    return 0;
}
