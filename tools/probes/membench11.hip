// membench11.hip -- does the bare 19-in / 19-out row-interleaved stream depend on the launch length?
// Copy and pull pattern (no arithmetic) over 256 x 256 x D cells, D = 256 and 1024, in one launch and
// in launches of K layers.  Buffers padded by one ghost row / layer on every side.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <initializer_list>
constexpr int W = 256, H = 256, HP = H + 2;
__constant__ int CX[19] = {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0};
__constant__ int CY[19] = {0, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 1, -1, 1, -1};
__constant__ int CZ[19] = {0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, -1, -1, 1, 1, -1, -1, 1};

template <int PULL> __global__ __launch_bounds__(256) void k_ri(const float* __restrict__ in, float* __restrict__ out, int z0) {
    const int x = threadIdx.x, y = blockIdx.x + 1, z = z0 + (int)blockIdx.y + 1;
    const long long own = ((long long)z * HP + y) * (19 * W) + x;
    float v[19];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        long long a = own + k * W;
        if (PULL) a = ((long long)(z - CZ[k]) * HP + (y - CY[k])) * (19 * W) + k * W + (x - CX[k]);
        v[k] = __builtin_nontemporal_load(in + a);
    }
#pragma unroll
    for (int k = 0; k < 19; k++) __builtin_nontemporal_store(v[k] * 1.0001f, out + own + k * W);
}

// One launch over the whole depth, but a workgroup of layer chunk c (K layers) does not start its
// loads before chunk c - LAG has (nearly) completed: a bounded spin on a counter, purely a pacing hint
// (the spin gives up after a fixed number of polls, so it cannot hang and changes no result).
template <int PULL> __global__ __launch_bounds__(256) void k_ri_window(const float* __restrict__ in, float* __restrict__ out,
                                                                       unsigned* __restrict__ done, int K, int lag, unsigned per_chunk) {
    const int x = threadIdx.x, y = blockIdx.x + 1, z = (int)blockIdx.y + 1;
    const int c = (int)blockIdx.y / K;
    if (c >= lag) {
        int polls = 0;
        // 64 counters per chunk, one per residue of the row index, 128 bytes apart: a single counter would
        // serialise 65 k atomics per launch at the memory side (measured: 2 x the kernel time)
        while (__hip_atomic_load(done + ((c - lag) * 64 + (blockIdx.x & 63)) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < per_chunk / 64 && ++polls < 2000)
            __builtin_amdgcn_s_sleep(8);
    }
    const long long own = ((long long)z * HP + y) * (19 * W) + x;
    float v[19];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        long long a = own + k * W;
        if (PULL) a = ((long long)(z - CZ[k]) * HP + (y - CY[k])) * (19 * W) + k * W + (x - CX[k]);
        v[k] = __builtin_nontemporal_load(in + a);
    }
#pragma unroll
    for (int k = 0; k < 19; k++) __builtin_nontemporal_store(v[k] * 1.0001f, out + own + k * W);
    if (threadIdx.x == 0) __hip_atomic_fetch_add(done + (c * 64 + (blockIdx.x & 63)) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
    for (int D : {256, 1024}) {
        const long long total = (long long)(D + 2) * HP * 19 * W + 4096;  // floats; every shifted read stays inside
        float *A, *B;
        if (hipMalloc(&A, total * 4) != hipSuccess || hipMalloc(&B, total * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
        (void)hipMemset(A, 0, total * 4); (void)hipMemset(B, 0, total * 4);
        const double bytes = 38.0 * W * H * D * 4;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int pull = 0; pull < 2; pull++)
            for (int K : {0, 512, 256, 128, 96, 64, 32, 16}) {
                if (K >= D) continue;
                const int chunk = K ? K : D;
                auto step = [&](const float* in, float* out) {
                    for (int z0 = 0; z0 < D; z0 += chunk) {
                        const int nl = D - z0 < chunk ? D - z0 : chunk;
                        if (pull) k_ri<1><<<dim3(H, nl), 256>>>(in, out, z0); else k_ri<0><<<dim3(H, nl), 256>>>(in, out, z0);
                    }
                };
                step(A, B); step(B, A); (void)hipDeviceSynchronize();
                float best = 1e9;
                for (int r = 0; r < 3; r++) {
                    (void)hipEventRecord(e0);
                    for (int i = 0; i < 4; i++) { step(A, B); step(B, A); }
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 8; best = ms < best ? ms : best;
                }
                printf("256x256x%-4d %s layers/launch %-4d : %8.1f us %6.3f TB/s\n", D, pull ? "pull" : "copy", chunk, best * 1e3, bytes / (best * 1e-3) / 1e12);
                fflush(stdout);
            }
        {   // soft window inside one launch
            unsigned* done; const size_t done_bytes = (size_t)128 * 64 * 32 * sizeof(unsigned); (void)hipMalloc(&done, done_bytes);
            for (int pull = 0; pull < 2; pull++)
                for (int K : {64, 32, 16, 8})
                    for (int lag : {1, 2, 4}) {
                        if (D / K > 128) continue;
                        auto step = [&](const float* in, float* out) {
                            (void)hipMemsetAsync(done, 0, done_bytes, 0);
                            if (pull) k_ri_window<1><<<dim3(H, D), 256>>>(in, out, done, K, lag, (unsigned)(H * K));
                            else k_ri_window<0><<<dim3(H, D), 256>>>(in, out, done, K, lag, (unsigned)(H * K));
                        };
                        step(A, B); step(B, A); (void)hipDeviceSynchronize();
                        float best = 1e9;
                        for (int r = 0; r < 3; r++) {
                            (void)hipEventRecord(e0);
                            for (int i = 0; i < 4; i++) { step(A, B); step(B, A); }
                            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                            float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 8; best = ms < best ? ms : best;
                        }
                        printf("256x256x%-4d %s one launch, window of %d chunks of %-3d layers : %8.1f us %6.3f TB/s\n", D, pull ? "pull" : "copy", lag, K, best * 1e3, bytes / (best * 1e-3) / 1e12);
                        fflush(stdout);
                    }
            (void)hipFree(done);
        }
        // the same chunks dealt alternately to two streams, joined once per step: the tail of one launch
        // overlaps the next launch of the other stream
        hipStream_t s0, s1; (void)hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
        hipEvent_t fork, join; (void)hipEventCreateWithFlags(&fork, hipEventDisableTiming); (void)hipEventCreateWithFlags(&join, hipEventDisableTiming);
        for (int pull = 0; pull < 2; pull++)
            for (int K : {128, 64, 32, 16, 8}) {
                if (K >= D) continue;
                auto step = [&](const float* in, float* out) {
                    (void)hipEventRecord(fork, s0); (void)hipStreamWaitEvent(s1, fork, 0);
                    int i = 0;
                    for (int z0 = 0; z0 < D; z0 += K, i++) {
                        const int nl = D - z0 < K ? D - z0 : K;
                        hipStream_t st = (i & 1) ? s1 : s0;
                        if (pull) k_ri<1><<<dim3(H, nl), 256, 0, st>>>(in, out, z0); else k_ri<0><<<dim3(H, nl), 256, 0, st>>>(in, out, z0);
                    }
                    (void)hipEventRecord(join, s1); (void)hipStreamWaitEvent(s0, join, 0);
                };
                step(A, B); step(B, A); (void)hipDeviceSynchronize();
                float best = 1e9;
                for (int r = 0; r < 3; r++) {
                    (void)hipEventRecord(e0, s0);
                    for (int i = 0; i < 4; i++) { step(A, B); step(B, A); }
                    (void)hipEventRecord(e1, s0); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 8; best = ms < best ? ms : best;
                }
                printf("256x256x%-4d %s layers/launch %-4d two streams : %8.1f us %6.3f TB/s\n", D, pull ? "pull" : "copy", K, best * 1e3, bytes / (best * 1e-3) / 1e12);
                fflush(stdout);
            }
        (void)hipStreamDestroy(s0); (void)hipStreamDestroy(s1);
        (void)hipFree(A); (void)hipFree(B);
    }
    return 0;
}
