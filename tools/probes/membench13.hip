// membench13.hip -- what makes the AA odd phase slow: the x-shifted stores or the scatter over nine rows?
// Bare patterns on the row-interleaved layout (256^3 + ghost shell), in place, launches of 64 layers:
//   even : read own slots, write own slots                      (AA even phase)
//   odd  : read slot (k, c - c_k), write slot (k, c + c_k)      (AA odd phase as implemented)
//   oddx : read slot (k, c - c_k), write slot (k, c + (0,cy,cz)) (same rows, stores aligned in x)
//   pull : read slot (k, c - c_k), write own slots              (A/B pattern, in place for reference)
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int W = 256, H = 256, D = 256, WP = 288, HP = H + 2;
__constant__ int CX[19] = {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0};
__constant__ int CY[19] = {0, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 1, -1, 1, -1};
__constant__ int CZ[19] = {0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, -1, -1, 1, 1, -1, -1, 1};

template <int MODE> __global__ __launch_bounds__(256) void k(float* M, int z0) {
    const int x = threadIdx.x, y = blockIdx.x + 1, z = z0 + (int)blockIdx.y + 1;
    const long long row = 19 * WP, layer = (long long)HP * row;
    const long long own = (long long)z * layer + (long long)y * row + x + 32;  // x = 0 on a 128-byte line, ghost slot in front
    float v[19];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        const long long d = (long long)CZ[k] * layer + (long long)CY[k] * row + CX[k];
        const float* p = M + own + k * WP - (MODE == 0 ? 0 : d);
        v[k] = (MODE != 0 && CX[k] != 0) ? *p : __builtin_nontemporal_load(p);
    }
#pragma unroll
    for (int k = 0; k < 19; k++) {
        const long long d = (long long)CZ[k] * layer + (long long)CY[k] * row + CX[k];
        const long long dyz = (long long)CZ[k] * layer + (long long)CY[k] * row;
        float* p = M + own + k * WP + (MODE == 1 ? d : (MODE == 2 ? dyz : 0));
        if (MODE == 1 && CX[k] != 0) *p = v[k] * 1.0001f;
        else __builtin_nontemporal_store(v[k] * 1.0001f, p);
    }
}

int main() {
    const long long total = (long long)(D + 2) * HP * 19 * WP + 8192;
    float* A;
    if (hipMalloc(&A, total * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(A, 0, total * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double bytes = 38.0 * W * H * D * 4;
    auto run = [&](const char* name, auto launch) {
        auto pass = [&] { for (int z0 = 0; z0 < D; z0 += 64) launch(z0); };
        pass(); (void)hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 3; r++) {
            (void)hipEventRecord(e0); for (int i = 0; i < 8; i++) pass(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 8; best = ms < best ? ms : best;
        }
        printf("%-6s : %7.1f us %6.3f TB/s\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12); fflush(stdout);
    };
    run("even", [&](int z0) { k<0><<<dim3(H, 64), 256>>>(A, z0); });
    run("odd", [&](int z0) { k<1><<<dim3(H, 64), 256>>>(A, z0); });
    run("oddx", [&](int z0) { k<2><<<dim3(H, 64), 256>>>(A, z0); });
    run("pull", [&](int z0) { k<3><<<dim3(H, 64), 256>>>(A, z0); });
    return 0;
}
