// k_fields.h -- derived fields: curl, strided velocity sample, speed.
// Part of the single translation unit lbm_hip.hip (included inside its anonymous namespace, after
// <hip/hip_runtime.h>, include/lbm_abi.h and d3q19.h); not a stand-alone header.
// |curl u| by clamped central differences (lbm_solver.cu:612-656).  One wave per 64-cell row segment marching up a
// chunk of layers: the column's three layers of u stay in registers (one new float4 per cell and layer from HBM),
// the x neighbours come from the neighbouring lanes (the two edge lanes load theirs), the y neighbours from cache.
// Same operands and operations as the reference's one-thread-per-cell form, so the values are the same bits.
constexpr int CURL_ZCHUNK = 32;
__global__ __launch_bounds__(256) void k_curl(const float4* __restrict__ u, float* __restrict__ curl, int W, int H, int D) {
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int z0 = blockIdx.z * CURL_ZCHUNK, z1 = min(D, z0 + CURL_ZCHUNK);
    if (y >= H) return;                                             // (whole waves)
    const int xc = min(x, W - 1);                                   // lanes beyond the row follow its last cell, store nothing
    const long long layer = (long long)H * W;
    const float4* col = u + (long long)y * W + xc;
    const long long up = (long long)(min(y + 1, H - 1) - y) * W, down = (long long)(max(y - 1, 0) - y) * W;
    const int xl = max(xc - 1, 0) - xc, xr = min(xc + 1, W - 1) - xc;
    const bool edge_l = lane == 0, edge_r = lane == 63 || x >= W - 1;
    float4 zm = col[(long long)max(z0 - 1, 0) * layer];
    float4 c = col[(long long)z0 * layer];
    // four layers per turn, every load of the turn issued before the first use: a wave that waited for each layer's
    // loads in turn spent its life on memory latency (32 layers x one round trip: 100 us at 256^3, no faster than
    // one thread per cell)
    constexpr int U = 4;
    for (int zb = z0; zb < z1; zb += U) {
        float4 zp[U], yp[U], ym[U], er[U], el[U];
#pragma unroll
        for (int i = 0; i < U; i++) {
            const int z = min(zb + i, z1 - 1);                      // (a ragged last turn repeats its last layer, stores nothing for it)
            const float4* here = col + (long long)z * layer;
            zp[i] = here[(long long)(min(z + 1, D - 1) - z) * layer];
            yp[i] = here[up];
            ym[i] = here[down];
            if (edge_r) er[i] = here[xr];
            if (edge_l) el[i] = here[xl];
        }
#pragma unroll
        for (int i = 0; i < U; i++) {
            const int z = zb + i;
            // x neighbours: the next / previous lane's own cell of this layer; the wave's edge lanes (and the row's last cell) load
            float xp_y = __shfl_down(c.y, 1), xp_z = __shfl_down(c.z, 1);
            float xm_y = __shfl_up(c.y, 1), xm_z = __shfl_up(c.z, 1);
            if (edge_r) { xp_y = er[i].y; xp_z = er[i].z; }
            if (edge_l) { xm_y = el[i].y; xm_z = el[i].z; }
            const float dv_dx = (xp_y - xm_y) * 0.5f;
            const float dw_dx = (xp_z - xm_z) * 0.5f;
            const float du_dy = (yp[i].x - ym[i].x) * 0.5f;
            const float dw_dy = (yp[i].z - ym[i].z) * 0.5f;
            const float du_dz = (zp[i].x - zm.x) * 0.5f;
            const float dv_dz = (zp[i].y - zm.y) * 0.5f;
            const float cxv = dw_dy - dv_dz;
            const float cyv = du_dz - dw_dx;
            const float czv = dv_dx - du_dy;
            if (x < W && z < z1) curl[(long long)z * layer + (long long)y * W + x] = sqrtf(cxv * cxv + cyv * cyv + czv * czv);
            zm = c;
            c = zp[i];
        }
    }
}

// Derived fields for the visualisation consumers either side of the path (the reference's renderer
// samples u as an RGBA32F volume: renderer.cpp:1382-1389, raymarch.frag / streamline.comp).
// strided sample of u: out[(k*ny + j)*nx + i] = u(i*sx, j*sy, k*sz); speed = |u.xyz|
__global__ void k_sample_velocity(const float4* __restrict__ u, float4* __restrict__ out, int W, int H,
                                  int nx, int ny, int nz, int sx, int sy, int sz) {
    const long long i = gidx();
    if (i >= (long long)nx * ny * nz) return;
    const int ix = (int)(i % nx), iy = (int)((i / nx) % ny), iz = (int)(i / ((long long)nx * ny));
    out[i] = u[((long long)iz * sz * H + (long long)iy * sy) * W + (long long)ix * sx];
}
__global__ void k_speed(const float4* __restrict__ u, float* __restrict__ out, long long n) {
    const long long i = gidx();
    if (i >= n) return;
    const float4 v = u[i];
    out[i] = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);  // cuda_math::length, cuda_math.cuh:142-145
}
