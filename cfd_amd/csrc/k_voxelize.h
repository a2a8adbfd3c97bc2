// k_voxelize.h -- device voxeliser.
// Part of the single translation unit lbm_hip.hip (included inside its anonymous namespace, after
// <hip/hip_runtime.h>, include/lbm_abi.h and d3q19.h); not a stand-alone header.
// ---- device voxeliser (MeshVoxelizer::voxelize, mesh_voxelizer.cpp:146-216, on the GPU) ----------
// One thread per +X ray (one (y,z) row of the grid), one block per 256 consecutive rows of a z
// layer.  Triangles stream through LDS in tiles of 256; the z-extent cull is uniform for the
// block, the y-extent cull and the Moeller-Trumbore test (mesh_voxelizer.cpp:89-121) run per ray
// in the reference's fp32 operation order (contraction off, IEEE division), so the hit distances
// -- and with them ceil/floor of the span ends -- equal the host's bit for bit.  Each thread sorts
// its hits, turns the pairs into bits and writes its row of the packed mask itself (whole words,
// no atomics); y = 0 becomes the solid floor (lbm_solver.cu:549-553).  A ray with more than
// VOX_MAX_HITS hits raises `overflow` and the caller falls back to the host voxeliser.
constexpr int VOX_MAX_HITS = 64;
constexpr int VOX_TILE = 256;
struct VoxTri {
    float v0x, v0y, v0z, v1x, v1y, v1z, v2x, v2y, v2z;
    float ymin, ymax, zmin, zmax;
};
struct VoxFrame {
    float gcx, gcy, gcz;  // grid centre (W/2, H/2, D/2)
    float mcx, mcy, mcz;  // mesh point mapped onto it
    float scale;
};
__device__ __forceinline__ bool vox_hit_plus_x(float ox, float oy, float oz, const VoxTri& t, float& dist) {
    constexpr float kEps = 0.0000001f;
    const float e1x = t.v1x - t.v0x, e1y = t.v1y - t.v0y, e1z = t.v1z - t.v0z;
    const float e2x = t.v2x - t.v0x, e2y = t.v2y - t.v0y, e2z = t.v2z - t.v0z;
    // h = cross(dir, e2) with dir = (1,0,0), written out as glm::cross evaluates it
    const float hx = 0.0f * e2z - e2y * 0.0f, hy = 0.0f * e2x - e2z * 1.0f, hz = 1.0f * e2y - e2x * 0.0f;
    const float a = e1x * hx + e1y * hy + e1z * hz;
    if (a > -kEps && a < kEps) return false;
    const float f = 1.0f / a;
    const float sx = ox - t.v0x, sy = oy - t.v0y, sz = oz - t.v0z;
    const float u = f * (sx * hx + sy * hy + sz * hz);
    if (u < 0.0f || u > 1.0f) return false;
    const float qx = sy * e1z - e1y * sz, qy = sz * e1x - e1z * sx, qz = sx * e1y - e1x * sy;
    const float v = f * (1.0f * qx + 0.0f * qy + 0.0f * qz);
    if (v < 0.0f || u + v > 1.0f) return false;
    dist = f * (e2x * qx + e2y * qy + e2z * qz);
    return dist > kEps;
}
__global__ __launch_bounds__(256) void k_voxelize_rows(const VoxTri* __restrict__ tris, int ntris, VoxFrame fr,
                                                       unsigned long long* __restrict__ mask, Geo g,
                                                       int* __restrict__ overflow) {
    __shared__ VoxTri tile[VOX_TILE];
    const int z = blockIdx.y;
    const int y = blockIdx.x * 256 + threadIdx.x;
    const float ox = (0.0f - fr.gcx) / fr.scale + fr.mcx;
    const float oy = ((float)y - fr.gcy) / fr.scale + fr.mcy;
    const float oz = ((float)(z + g.z0) - fr.gcz) / fr.scale + fr.mcz;  // z is this context's layer; the grid is the global domain
    float hits[VOX_MAX_HITS];
    int nh = 0;
    bool over = false;
    for (int base = 0; base < ntris; base += VOX_TILE) {
        const int cnt = min(VOX_TILE, ntris - base);
        __syncthreads();
        if ((int)threadIdx.x < cnt) tile[threadIdx.x] = tris[base + threadIdx.x];
        __syncthreads();
        if (y >= g.H) continue;
        for (int i = 0; i < cnt; i++) {
            const VoxTri& t = tile[i];
            if (oz < t.zmin || oz > t.zmax) continue;  // uniform for the block
            if (oy < t.ymin || oy > t.ymax) continue;
            float d;
            if (vox_hit_plus_x(ox, oy, oz, t, d)) {
                if (nh < VOX_MAX_HITS) hits[nh++] = d; else over = true;
            }
        }
    }
    if (y >= g.H) return;
    if (over) { *overflow = 1; return; }
    for (int i = 1; i < nh; i++) {  // insertion sort (std::sort on the host: same order of values)
        const float v = hits[i];
        int j = i - 1;
        while (j >= 0 && hits[j] > v) { hits[j + 1] = hits[j]; j--; }
        hits[j + 1] = v;
    }
    unsigned long long* row = mask + ((long long)(z + 1) * g.H + y) * g.Wm;
    for (int w = 0; w < g.Wm; w++) {
        const int lo = w * 64;
        const int valid = min(64, g.W - lo);
        const unsigned long long all = valid >= 64 ? ~0ull : ((1ull << valid) - 1ull);
        unsigned long long bits = 0;
        if (y == 0) {
            bits = all;  // floor
        } else {
            for (int i = 0; i + 1 < nh; i += 2) {
                const int x0 = max(0, (int)ceilf(hits[i] * fr.scale));
                const int x1 = min(g.W, (int)floorf(hits[i + 1] * fr.scale));
                const int a = max(x0, lo) - lo, b = min(x1, lo + 64) - lo;  // [a, b) inside this word
                if (b > a) bits |= (b - a >= 64 ? ~0ull : ((1ull << (b - a)) - 1ull)) << a;
            }
        }
        row[w] = bits;
    }
}

// per-tile path flag: 0 = interior fast path, 1 = full boundary chain, 2 = every cell solid (skipped)
