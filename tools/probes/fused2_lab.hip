// fused2_lab.hip -- tuning lab for the two-steps-per-pass kernel (round 2).  NOT product: interior cells of
// a fixed 256^3 row-interleaved lattice only, like tools/probes/fused2_proto.hip; it exists to pick the shape of
// cfd_amd/csrc/k_fused2.h by measurement.  One templated kernel covers the variants:
//   TY     rows per tile (TY + 2 row waves + 1 halo-column wave per workgroup)
//   STAGE  0: the 19 loads of a layer are issued when they are needed (no prefetch)
//          1: the NEXT layer's loads go global -> LDS (global_load_lds_dword) into a wave-private area
//          2: the NEXT layer's loads go into registers
//   NBUF   1: one exchange buffer, two barriers per layer;  2: two buffers, one barrier per layer
//   MINW   second __launch_bounds__ argument (waves per SIMD the compiler must allow)
//   STAMP  1: every wave accumulates s_memtime deltas per phase (printed for a few workgroups)
// Checked bit for bit against two plain steps on every cell the tiles cover.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../cfd_amd/csrc/d3q19.h"
using namespace d3q19;

constexpr int W = 256, H = 256, D = 256;
constexpr int WP = W + 64, XOFF = 32;
constexpr long long ROW = 19ll * WP, LAYER = (long long)H * ROW, TOTAL = (long long)D * LAYER;
constexpr int ZS = 2, ZE = D - 2;
constexpr float OMEGA_S = 1.98f, OMEGA_B = 1.90f;
constexpr int LCOLS = 66;
constexpr int NPH = 8;

__device__ __forceinline__ long long at(int k, int x, int y, int z) {
    return (long long)z * LAYER + (long long)y * ROW + (long long)k * WP + x + XOFF;
}

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

__host__ __device__ constexpr int lds_plane(int k) {
    return k == 11 ? 0 : k == 14 ? 1 : k == 15 ? 2 : k == 18 ? 3 :
           k == 1 ? 4 : k == 2 ? 5 : k == 3 ? 6 : k == 4 ? 7 : k == 7 ? 8 : k == 8 ? 9 : k == 9 ? 10 : k == 10 ? 11 :
           k == 12 ? 12 : k == 13 ? 13 : k == 16 ? 14 : k == 17 ? 15 : -1;
}

template <int K, int PLANE> struct Put {
    static __device__ __forceinline__ void run(float* lds, int slot, const float* up_prev, const float* cur, const float* nxt) {
        Put<K - 1, PLANE>::run(lds, slot, up_prev, cur, nxt);
        constexpr int p = lds_plane(K);
        if (p >= 0) lds[p * PLANE + slot] = CZ[K] == 1 ? up_prev[K] : (CZ[K] == 0 ? cur[K] : nxt[K]);
    }
};
template <int PLANE> struct Put<-1, PLANE> { static __device__ __forceinline__ void run(float*, int, const float*, const float*, const float*) {} };

template <int K, int PLANE> struct Get {
    static __device__ __forceinline__ void run(const float* lds, int r, int col, const float* up_prev, const float* cur, const float* nxt, float* f) {
        Get<K - 1, PLANE>::run(lds, r, col, up_prev, cur, nxt, f);
        constexpr int p = lds_plane(K);
        if (p >= 0) f[K] = lds[p * PLANE + (r - CY[K]) * LCOLS + (col - CX[K])];
        else f[K] = K == 0 ? cur[0] : (K == 5 ? up_prev[5] : nxt[6]);
    }
};
template <int PLANE> struct Get<-1, PLANE> { static __device__ __forceinline__ void run(const float*, int, int, const float*, const float*, const float*, float*) {} };

#define STAMP_AT(i)                                                      \
    if (STAMP) {                                                         \
        const unsigned long long now_ = __builtin_readcyclecounter();    \
        acc[i] += now_ - tprev;                                          \
        tprev = now_;                                                    \
    }

template <int TY, int STAGE, int NBUF, int MINW, int STAMP>
__global__ __launch_bounds__((TY + 3) * 64, MINW) void k_lab(const float* __restrict__ in, float* __restrict__ out, int zchunk,
                                                            unsigned long long* __restrict__ stamps) {
    constexpr int ROWS = TY + 2;
    constexpr int PLANE = ROWS * LCOLS;
    constexpr int EXCH = 16 * PLANE;
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * 64, y0 = 2 + blockIdx.y * TY;
    const int zs = ZS + blockIdx.z * zchunk, ze = min(zs + zchunk, ZE);
    int x, r;
    bool active = true;
    if (wave < ROWS) { r = wave; x = x0 + lane; }
    else { r = lane % ROWS; x = lane < ROWS ? x0 - 1 : x0 + 64; active = lane < 2 * ROWS; }
    const int y = y0 - 1 + r;
    const int col = x - x0 + 1;
    const int slot = r * LCOLS + col;
    const bool second = wave >= 1 && wave <= TY;
    float* stage = lds + NBUF * EXCH + wave * (19 * 64);

    unsigned long long acc[NPH];
    unsigned long long tprev = 0;
    if (STAMP) {
#pragma unroll
        for (int i = 0; i < NPH; i++) acc[i] = 0;
        tprev = __builtin_readcyclecounter();
    }

    auto issue_lds = [&](int L) {
        if (active) {
#pragma unroll
            for (int k = 0; k < Q; k++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + at(k, x - CX[k], y - CY[k], L - CZ[k])),
                                                 (__attribute__((address_space(3))) void*)(stage + k * 64), 4, 0, 0);
        }
    };
    float up_prev[Q], cur[Q], nxt[Q], pf[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) { up_prev[k] = 0.f; cur[k] = 0.f; nxt[k] = 0.f; pf[k] = 0.f; }
    if (STAGE == 1) { issue_lds(zs - 1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    if (STAGE == 2 && active) {
#pragma unroll
        for (int k = 0; k < Q; k++) pf[k] = __builtin_nontemporal_load(in + at(k, x - CX[k], y - CY[k], zs - 1 - CZ[k]));
    }
    int buf = 0;
    for (int L = zs - 1; L <= ze; L++) {
        // ---- first step of layer L --------------------------------------------------------------
        if (active) {
            if (STAGE == 0) {
#pragma unroll
                for (int k = 0; k < Q; k++) pf[k] = __builtin_nontemporal_load(in + at(k, x - CX[k], y - CY[k], L - CZ[k]));
            } else if (STAGE == 1) {
#pragma unroll
                for (int k = 0; k < Q; k++) pf[k] = stage[k * 64 + lane];
            }
            const Moments m = moments_of(pf);
            collide_cell(pf, m, OMEGA_S, OMEGA_B, nxt);
        }
        STAMP_AT(0)
        if (STAGE == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging area has been read
            if (L < ze) issue_lds(L + 1);
        } else if (STAGE == 2) {
            if (active && L < ze) {
#pragma unroll
                for (int k = 0; k < Q; k++) pf[k] = __builtin_nontemporal_load(in + at(k, x - CX[k], y - CY[k], L + 1 - CZ[k]));
            }
        }
        STAMP_AT(1)
        // ---- second step of layer L - 1 -----------------------------------------------------------
        if (L >= zs + 1) {
            const int z = L - 1;
            float* my = lds + buf * EXCH;
            if (active) Put<Q - 1, PLANE>::run(my, slot, up_prev, cur, nxt);
            STAMP_AT(2)
            __syncthreads();
            STAMP_AT(3)
            if (second) {
                float f[Q], o[Q];
                Get<Q - 1, PLANE>::run(my, r, col, up_prev, cur, nxt, f);
                const Moments m = moments_of(f);
                collide_cell(f, m, OMEGA_S, OMEGA_B, o);
                STAMP_AT(4)
                if (STAGE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                STAMP_AT(5)
#pragma unroll
                for (int k = 0; k < Q; k++) __builtin_nontemporal_store(o[k], out + at(k, x, y, z));
                STAMP_AT(6)
            } else {
                if (STAGE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                STAMP_AT(5)
            }
            if (NBUF == 1) __syncthreads(); else buf ^= 1;
            STAMP_AT(7)
        } else {
            if (STAGE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP_AT(5)
        }
#pragma unroll
        for (int k = 0; k < Q; k++) { up_prev[k] = cur[k]; cur[k] = nxt[k]; }
    }
    if (STAMP && lane == 0) {
        const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
#pragma unroll
        for (int i = 0; i < NPH; i++) stamps[((long long)wg * 16 + wave) * NPH + i] = acc[i];
    }
}

static unsigned hash32(unsigned v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; }

static float *A, *B, *T, *R;
static unsigned long long* d_stamps;
static std::vector<float> h_ref;
static hipEvent_t e0, e1;

template <class F>
static float timeit(F f, int reps) {
    f(); (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0); for (int i = 0; i < reps; i++) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms / reps < best ? ms / reps : best;
    }
    return best;
}

template <int TY, int STAGE, int NBUF, int MINW>
static int run_variant(const char* name, const int* zsplits, int nz, bool stamps) {
    constexpr int ROWS = TY + 2, NT_Y = (H - 4) / TY;
    constexpr int threads = (TY + 3) * 64;
    const size_t bytes = (size_t)TOTAL * sizeof(float);
    const size_t lds_bytes = (size_t)(NBUF * 16 * ROWS * LCOLS + (STAGE == 1 ? (ROWS + 1) * 19 * 64 : 0)) * sizeof(float);
    auto kern = k_lab<TY, STAGE, NBUF, MINW, 0>;
    auto kern_s = k_lab<TY, STAGE, NBUF, MINW, 1>;
    if (lds_bytes > 163840 ||
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess ||
        hipFuncSetAttribute((const void*)kern_s, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
        printf("== %s: LDS %zu bytes refused\n", name, lds_bytes);
        (void)hipGetLastError();
        return 0;
    }
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void*)kern);
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, threads, lds_bytes);
    printf("== %s: TY %d, STAGE %d, NBUF %d, %d threads, LDS %zu B, %d VGPRs, scratch %zu B, %d workgroup(s)/CU\n", name, TY, STAGE, NBUF,
           threads, lds_bytes, fa.numRegs, (size_t)fa.localSizeBytes, occ);
    if (occ < 1) return 0;
    // correctness once, with the first z split
    {
        const int zsplit = zsplits[0], zchunk = (ZE - ZS + zsplit - 1) / zsplit;
        (void)hipMemset(B, 0, bytes);
        kern<<<dim3(W / 64, NT_Y, zsplit), threads, lds_bytes>>>(A, B, zchunk, d_stamps);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        std::vector<float> hb((size_t)TOTAL);
        (void)hipMemcpy(hb.data(), B, bytes, hipMemcpyDeviceToHost);
        long long bad = 0;
        for (int z = ZS; z < ZE; z++)
            for (int y = 2; y < 2 + NT_Y * TY; y++)
                for (int k = 0; k < 19; k++) {
                    const size_t i0 = (size_t)z * LAYER + (size_t)y * ROW + (size_t)k * WP + XOFF;
                    if (memcmp(&hb[i0 + 2], &h_ref[i0 + 2], (W - 4) * 4) != 0) bad++;
                }
        printf("   check: %lld rows differ from two plain steps\n", bad);
        if (bad) return 1;
    }
    const double cells_f = (double)W * NT_Y * TY * (ZE - ZS);
    for (int i = 0; i < nz; i++) {
        const int zsplit = zsplits[i], zchunk = (ZE - ZS + zsplit - 1) / zsplit;
        const dim3 gf(W / 64, NT_Y, zsplit);
        const float t = timeit([&] { kern<<<gf, threads, lds_bytes>>>(A, B, zchunk, d_stamps); }, 10);
        printf("   z split %2d (%4d workgroups, %3d layers each): %8.1f us  %7.0f MLUPS\n", zsplit, gf.x * gf.y * gf.z, zchunk, t * 1e3,
               2 * cells_f / (t * 1e-3) / 1e6);
    }
    if (stamps) {
        const int zsplit = zsplits[0], zchunk = (ZE - ZS + zsplit - 1) / zsplit;
        const dim3 gf(W / 64, NT_Y, zsplit);
        const int nwg = gf.x * gf.y * gf.z;
        (void)hipMemset(d_stamps, 0, (size_t)nwg * 16 * NPH * 8);
        const float t = timeit([&] { kern_s<<<gf, threads, lds_bytes>>>(A, B, zchunk, d_stamps); }, 3);
        std::vector<unsigned long long> hs((size_t)nwg * 16 * NPH);
        (void)hipMemcpy(hs.data(), d_stamps, hs.size() * 8, hipMemcpyDeviceToHost);
        printf("   stamped run: %8.1f us.  s_memtime ticks per layer and phase, mean over all workgroups\n", t * 1e3);
        printf("   phases: 0 stage-read+collide1 | 1 issue next loads | 2 put | 3 barrier | 4 get+collide2 | 5 wait loads | 6 stores | 7 end barrier/flip\n");
        const int show[4] = {0, 1, ROWS / 2, ROWS};
        for (int s = 0; s < 4; s++) {
            const int wv = show[s];
            double ph[NPH] = {0};
            for (int wg = 0; wg < nwg; wg++)
                for (int i = 0; i < NPH; i++) ph[i] += (double)hs[((size_t)wg * 16 + wv) * NPH + i];
            double tot = 0;
            for (int i = 0; i < NPH; i++) { ph[i] /= (double)nwg * (zchunk + 2); tot += ph[i]; }
            printf("   wave %2d:", wv);
            for (int i = 0; i < NPH; i++) printf(" %7.0f", ph[i]);
            printf("  | total %7.0f\n", tot);
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)TOTAL * sizeof(float);
    if (hipMalloc(&A, bytes) != hipSuccess || hipMalloc(&B, bytes) != hipSuccess || hipMalloc(&T, bytes) != hipSuccess ||
        hipMalloc(&R, bytes) != hipSuccess || hipMalloc(&d_stamps, (size_t)4096 * 16 * NPH * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    {
        std::vector<float> h((size_t)TOTAL);
        for (long long i = 0; i < TOTAL; i++) {
            const int k = (int)((i / WP) % 19);
            h[(size_t)i] = weight(k) * (1.0f + 0.02f * ((hash32((unsigned)i) & 0xffff) / 65535.0f - 0.5f));
        }
        (void)hipMemcpy(A, h.data(), bytes, hipMemcpyHostToDevice);
    }
    (void)hipMemset(B, 0, bytes); (void)hipMemset(T, 0, bytes); (void)hipMemset(R, 0, bytes);
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const dim3 gref(H - 2, D - 2);
    k_step1<<<gref, 256>>>(A, T);
    k_step1<<<gref, 256>>>(T, R);
    if (hipDeviceSynchronize() != hipSuccess) { printf("reference failed\n"); return 1; }
    h_ref.resize((size_t)TOTAL);
    (void)hipMemcpy(h_ref.data(), R, bytes, hipMemcpyDeviceToHost);
    const float t1 = timeit([&] { k_step1<<<gref, 256>>>(A, T); k_step1<<<gref, 256>>>(T, R); }, 10);
    printf("two plain steps : %8.1f us  %7.0f MLUPS\n", t1 * 1e3, 2.0 * W * (H - 2) * (D - 2) / (t1 * 1e-3) / 1e6);
    const bool st = argc > 1 && atoi(argv[1]) != 0;
    int rc = 0;
    { const int zs[] = {3, 6};     rc |= run_variant<12, 1, 1, 1>("C   (LDS-staged prefetch, 2 barriers)", zs, 2, st); }
    { const int zs[] = {3, 6};     rc |= run_variant<12, 0, 2, 1>("A   (no prefetch, 1 barrier)", zs, 2, st); }
    { const int zs[] = {2, 4, 7};  rc |= run_variant<9, 2, 2, 1>("D9r (register prefetch, 1 barrier)", zs, 3, st); }
    { const int zs[] = {2, 4, 7};  rc |= run_variant<9, 1, 2, 1>("D9s (LDS-staged prefetch, 1 barrier)", zs, 3, st); }
    { const int zs[] = {2, 4, 7};  rc |= run_variant<9, 0, 2, 1>("D9n (no prefetch, 1 barrier)", zs, 3, st); }
    { const int zs[] = {3, 6};     rc |= run_variant<6, 1, 1, 5>("E6s (LDS-staged, 2 barriers, 2 workgroups/CU)", zs, 2, st); }
    { const int zs[] = {3, 6};     rc |= run_variant<6, 0, 1, 5>("E6n (no prefetch, 2 barriers, 2 workgroups/CU)", zs, 2, st); }
    { const int zs[] = {2, 4};     rc |= run_variant<4, 1, 2, 4>("F4s (LDS-staged, 1 barrier, 2+ workgroups/CU)", zs, 2, st); }
    return rc;
}
