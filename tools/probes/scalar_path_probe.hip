// Does the scalar data cache's read path add to what a CU can keep in flight?  k_fused2 is bound by the line requests its
// vector L1 holds in flight (profiles/r03_tuning.md section 6); 42 % of its read requests are partial lines that carry 8-16
// useful bytes (the halo columns).  Here every wave streams full lines through the vector path (mode bit 0) and, beside
// them, 8-byte pieces of other lines either through the vector path too (bit 2: eight lanes, eight lines, one instruction)
// or through scalar loads (bit 1: s_load_dwordx2, eight in flight per wave).  Reads only; nothing is written through the
// scalar cache.
// build: hipcc -O3 --offload-arch=gfx950 -o scalar_path_probe scalar_path_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ITERS = 384;
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(1024) void k(const u4* __restrict__ full, const char* __restrict__ part, unsigned* out, int mode) {
    const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 16u + (threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    // full lines: 1 KB per wave and iteration (eight 128-byte lines, one 16-byte load per lane)
    const u4* f = full + (size_t)wave * ITERS * 64 + lane;
    // partial lines: eight lines per wave and iteration, 8 bytes of each
    const char* p = part + (size_t)wave * ITERS * 1024;
    unsigned acc = 0;
    for (int i = 0; i < ITERS; i++) {
        u4 v = {0, 0, 0, 0};
        if (mode & 1) v = __builtin_nontemporal_load(f + (size_t)i * 64);
        if (mode & 2) {
            const char* q = p + (size_t)i * 1024;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)q), hi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)q >> 32));
            const unsigned long long a = ((unsigned long long)hi << 32) | lo;
            unsigned long long s0, s1, s2, s3, s4, s5, s6, s7;
            asm volatile("s_load_dwordx2 %0, %8, 0x0\n\t"
                         "s_load_dwordx2 %1, %8, 0x80\n\t"
                         "s_load_dwordx2 %2, %8, 0x100\n\t"
                         "s_load_dwordx2 %3, %8, 0x180\n\t"
                         "s_load_dwordx2 %4, %8, 0x200\n\t"
                         "s_load_dwordx2 %5, %8, 0x280\n\t"
                         "s_load_dwordx2 %6, %8, 0x300\n\t"
                         "s_load_dwordx2 %7, %8, 0x380\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "=&s"(s4), "=&s"(s5), "=&s"(s6), "=&s"(s7)
                         : "s"(a) : "memory");
            acc ^= (unsigned)(s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7);
        }
        if (mode & 4) {
            if (lane < 8) {
                const u2 w = __builtin_nontemporal_load(reinterpret_cast<const u2*>(p + (size_t)i * 1024 + lane * 128));
                acc ^= w.x ^ w.y;
            }
        }
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[wave * 64 + lane] = acc;   // (never: keeps the loads alive)
}

int main() {
    const size_t waves = 256 * 16;
    const size_t full_b = waves * ITERS * 1024, part_b = waves * ITERS * 1024;
    u4* full; char* part; unsigned* out;
    CHK(hipMalloc(&full, full_b)); CHK(hipMalloc(&part, part_b)); CHK(hipMalloc(&out, waves * 64 * 4));
    CHK(hipMemset(full, 1, full_b)); CHK(hipMemset(part, 2, part_b));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const char* names[8] = {"", "full lines on the vector path", "8-byte pieces on the scalar path", "full lines + pieces on the scalar path",
                            "8-byte pieces on the vector path", "full lines + pieces on the vector path", "", ""};
    for (int mode : {1, 4, 2, 5, 3}) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, full, part, out, mode);
            CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double lines_full = (mode & 1) ? (double)waves * ITERS * 8 : 0, pieces = (mode & 6) ? (double)waves * ITERS * 8 : 0;
        printf("mode %d (%s): %.1f us; full lines %.2f TB/s; pieces %.1f M requests = %.0f per CU and us\n", mode, names[mode], best * 1e3,
               lines_full * 128 / (best * 1e-3) / 1e12, pieces / 1e6, pieces / 256 / (best * 1e3));
    }
    return 0;
}
