// k_fields.h -- derived fields: curl, strided velocity sample, speed.
// Part of the single translation unit lbm_hip.hip (included inside its anonymous namespace, after
// <hip/hip_runtime.h>, include/lbm_abi.h and d3q19.h); not a stand-alone header.
__global__ void k_curl(const float4* __restrict__ u, float* __restrict__ curl, int W, int H, int D) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int z = blockIdx.z;
    if (x >= W) return;
    auto at = [&](int ix, int iy, int iz) -> float4 {
        ix = max(0, min(ix, W - 1));
        iy = max(0, min(iy, H - 1));
        iz = max(0, min(iz, D - 1));
        return u[((long long)iz * H + iy) * W + ix];
    };
    const float4 xp = at(x + 1, y, z), xm = at(x - 1, y, z);
    const float4 yp = at(x, y + 1, z), ym = at(x, y - 1, z);
    const float4 zp = at(x, y, z + 1), zm = at(x, y, z - 1);
    const float dv_dx = (xp.y - xm.y) * 0.5f;
    const float dw_dx = (xp.z - xm.z) * 0.5f;
    const float du_dy = (yp.x - ym.x) * 0.5f;
    const float dw_dy = (yp.z - ym.z) * 0.5f;
    const float du_dz = (zp.x - zm.x) * 0.5f;
    const float dv_dz = (zp.y - zm.y) * 0.5f;
    const float cxv = dw_dy - dv_dz;
    const float cyv = du_dz - dw_dx;
    const float czv = dv_dx - du_dy;
    curl[((long long)z * H + y) * W + x] = sqrtf(cxv * cxv + cyv * cyv + czv * czv);
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
