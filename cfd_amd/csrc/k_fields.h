// k_fields.h -- derived fields: curl, strided velocity sample, speed.
// Part of the single translation unit lbm_hip.hip (included inside its anonymous namespace, after
// <hip/hip_runtime.h>, include/lbm_abi.h and d3q19.h); not a stand-alone header.
// |curl u| by clamped central differences (lbm_solver.cu:612-656).  A workgroup of NW waves takes a 64 x (NW R) tile of
// rows up a chunk of layers, every wave R consecutive rows and U layers per turn: one new float4 per cell and layer
// from HBM, the column's three layers in registers, the x neighbours from the neighbouring lanes (the two edge lanes
// load theirs).  Inside a wave the rows above and below are its own registers; across waves the first and last row of
// every wave go through LDS, and the two rows outside the tile are loaded by the first and last wave.  Rows beyond the
// domain are loaded as copies of row H-1, which is what the reference's clamp reads.  Same operands and operations as
// the reference's one-thread-per-cell form, so the values are the same bits.
// (Round 3: one wave per row with the y neighbours left to the caches fetched 1.52 x the field from HBM, 98 us at
// 256^3 -- neighbouring workgroups drift apart by more than the L2 holds; this form: 70 us.)
// The grid is one-dimensional and dealt by hand: consecutive workgroup ids go round the 8 XCDs, so XCD k takes the
// k-th eighth of the tiles in (y band, z chunk, x tile) order.
constexpr int CURL_NW = 8, CURL_R = 2, CURL_U = 4;
template <int NW, int R, int U>
__global__ __launch_bounds__(NW * 64) void k_curl(const float4* __restrict__ u, float* __restrict__ curl, int W, int H, int D, int zchunk) {
    __shared__ float2 sh_first[U][NW + 1][64];      // [.][w]: first row of wave w; [.][NW]: the row above the tile
    __shared__ float2 sh_last[U][NW + 1][64];       // [.][w + 1]: last row of wave w; [.][0]: the row below the tile
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int RB = NW * R;
    const int gx = (W + 63) / 64, gz = (D + zchunk - 1) / zchunk;
    const int nb = (int)gridDim.x, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int L = xcd * (nb >> 3) + min(xcd, nb & 7) + slot;
    const int x = (L % gx) * 64 + lane;
    const int y0 = (L / (gx * gz)) * RB;
    const int z0 = ((L / gx) % gz) * zchunk, z1 = min(D, z0 + zchunk);
    const int xc = min(x, W - 1);
    const long long layer = (long long)H * W;
    const int xl = max(xc - 1, 0) - xc, xr = min(xc + 1, W - 1) - xc;
    const bool edge_l = lane == 0, edge_r = lane == 63 || x >= W - 1;
    const float4* col[R];
    int yr[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        yr[r] = y0 + w * R + r;
        col[r] = u + (long long)min(yr[r], H - 1) * W + xc;
    }
    const float4* halo = u + (long long)(w == 0 ? max(y0 - 1, 0) : min(y0 + RB, H - 1)) * W + xc;
    const bool halo_wave = w == 0 || w == NW - 1;
    float4 zm[R], c[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        zm[r] = col[r][(long long)max(z0 - 1, 0) * layer];
        c[r] = col[r][(long long)z0 * layer];
    }
    for (int zb = z0; zb < z1; zb += U) {
        float4 n[R][U];
        float2 er[R][U], el[R][U], hv[U];
#pragma unroll
        for (int i = 0; i < U; i++) {
            const int z = min(zb + i, z1 - 1);
            const long long zo = (long long)z * layer, zpo = (long long)min(z + 1, D - 1) * layer;
#pragma unroll
            for (int r = 0; r < R; r++) {
                n[r][i] = col[r][zpo];
                if (edge_r) { const float4* q = col[r] + zo + xr; er[r][i] = make_float2(q->y, q->z); }
                if (edge_l) { const float4* q = col[r] + zo + xl; el[r][i] = make_float2(q->y, q->z); }
            }
            if (halo_wave) { const float4 h = halo[zo]; hv[i] = make_float2(h.x, h.z); }
        }
        // the turn's centre layers: c, then the layers just loaded
#pragma unroll
        for (int i = 0; i < U; i++) {
            const float4 f = i == 0 ? c[0] : n[0][i - 1];
            const float4 l = i == 0 ? c[R - 1] : n[R - 1][i - 1];
            sh_first[i][w][lane] = make_float2(f.x, f.z);
            sh_last[i][w + 1][lane] = make_float2(l.x, l.z);
            if (w == 0) sh_last[i][0][lane] = hv[i];
            if (w == NW - 1) sh_first[i][NW][lane] = hv[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < U; i++) {
            const int z = zb + i;
            const float2 below = sh_last[i][w][lane], above = sh_first[i][w + 1][lane];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float4 cc = c[r];
                float xp_y = __shfl_down(cc.y, 1), xp_z = __shfl_down(cc.z, 1);
                float xm_y = __shfl_up(cc.y, 1), xm_z = __shfl_up(cc.z, 1);
                if (edge_r) { xp_y = er[r][i].x; xp_z = er[r][i].y; }
                if (edge_l) { xm_y = el[r][i].x; xm_z = el[r][i].y; }
                const float yp_x = r + 1 < R ? c[r + 1 < R ? r + 1 : r].x : above.x, yp_z = r + 1 < R ? c[r + 1 < R ? r + 1 : r].z : above.y;
                const float ym_x = r > 0 ? c[r > 0 ? r - 1 : 0].x : below.x, ym_z = r > 0 ? c[r > 0 ? r - 1 : 0].z : below.y;
                const float dv_dx = (xp_y - xm_y) * 0.5f;
                const float dw_dx = (xp_z - xm_z) * 0.5f;
                const float du_dy = (yp_x - ym_x) * 0.5f;
                const float dw_dy = (yp_z - ym_z) * 0.5f;
                const float du_dz = (n[r][i].x - zm[r].x) * 0.5f;
                const float dv_dz = (n[r][i].y - zm[r].y) * 0.5f;
                const float cxv = dw_dy - dv_dz;
                const float cyv = du_dz - dw_dx;
                const float czv = dv_dx - du_dy;
                if (x < W && z < z1 && yr[r] < H) curl[(long long)z * layer + (long long)yr[r] * W + x] = sqrtf(cxv * cxv + cyv * cyv + czv * czv);
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                zm[r] = c[r];
                c[r] = n[r][i];
            }
        }
        __syncthreads();
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
