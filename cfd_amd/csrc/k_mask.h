// k_mask.h -- tile flags, frozen solid cells, boundary classification read-back.
// Part of the single translation unit lbm_hip.hip (included inside its anonymous namespace, after
// <hip/hip_runtime.h>, include/lbm_abi.h and d3q19.h); not a stand-alone header.
template <int PRESET>
__global__ void k_tile_flags(const unsigned long long* __restrict__ mask, unsigned char* __restrict__ flags, Geo g) {
    const long long ti = gidx();
    const long long ntiles = (long long)g.D * g.H * g.Wt;
    if (ti >= ntiles) return;
    const long long row = ti / g.Wt;
    const int xs = (int)(ti - row * g.Wt);
    const int z = (int)(row / g.H);
    const int y = (int)(row - (long long)z * g.H);
    const int zg = z + g.z0;
    unsigned char fl = 0;
    const int cnt = min(64, g.W - xs * 64);
    const unsigned long long all = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
    // 0: interior tile, nothing solid within reach (the kernels' fast path); 1: a solid cell within reach; 2: all solid;
    // 3: on a wall of the domain, nothing solid within reach (the one-step kernels treat it like 1; the two-step
    // planner tells walls from obstacles by it)
    if ((mask[((long long)(z + 1) * g.H + y) * g.Wm + xs] & all) == all) {
        fl = 2;
    } else {
        unsigned long long any = 0;
        for (int dz = -1; dz <= 1; dz++)
            for (int dy = -1; dy <= 1; dy++) {
                if (y + dy < 0 || y + dy >= g.H) continue;     // (the ghost layers of the mask always exist)
                const unsigned long long* r = mask + ((long long)(z + 1 + dz) * g.H + (y + dy)) * g.Wm;
                any |= r[xs];
                if (xs > 0) any |= r[xs - 1] >> 63;
                if (xs + 1 < g.Wm) any |= r[xs + 1] & 1ull;
            }
        const bool wall = y == 0 || y == g.H - 1 || zg == 0 || zg == g.Dg - 1;
        fl = any != 0 ? 1 : (wall ? 3 : 0);
    }
    flags[ti] = fl;
}

// Solid cells keep their populations frozen; the reference re-copies them every step
// (lbm_solver.cu:263-265), which only matters because the outlet branch may read a solid cell.
// One copy cur -> other after the mask (or the populations) changed gives the same state.
__global__ void k_sync_solid(const float* __restrict__ cur, float* __restrict__ other,
                             const unsigned long long* __restrict__ mask, Geo g) {
    const long long i = gidx();
    const long long n = (long long)g.W * g.H * g.D;
    if (i >= n) return;
    const long long row = i / g.W;
    const int x = (int)(i - row * g.W);
    const int z = (int)(row / g.H);
    const int y = (int)(row - (long long)z * g.H);
    if (!solid_at(g, mask, x, y, z + 1)) return;
    const long long off = cell_off(g, x, y, z + 1);
    for (int k = 0; k < Q; k++) other[(long long)k * g.plane + off] = cur[(long long)k * g.plane + off];
}

template <int PRESET>
__global__ void k_classify(const unsigned long long* __restrict__ mask, unsigned char* __restrict__ cls, Geo g) {
    const long long i = gidx();
    const long long n = (long long)g.W * g.H * g.D;
    if (i >= n) return;
    const long long row = i / g.W;
    const int x = (int)(i - row * g.W);
    const int z = (int)(row / g.H);
    const int y = (int)(row - (long long)z * g.H);
    unsigned char out[Q];
    if (solid_at(g, mask, x, y, z + 1)) {
        for (int k = 0; k < Q; k++) out[k] = SRC_PULL;
    } else {
        GatherAll<PRESET, Q - 1>::classify(g, mask, x, y, z + 1, out);
    }
    for (int k = 0; k < Q; k++) cls[i * Q + k] = out[k];
}
