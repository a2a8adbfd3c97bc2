// k_aa.h -- the in-place AA-pattern variant of the step kernel and its helpers.
// Part of the single translation unit lbm_hip.hip (included inside its anonymous namespace, after
// <hip/hip_runtime.h>, include/lbm_abi.h and d3q19.h); not a stand-alone header.
// ------------------------------------------------------------------------------------------
// AA-pattern variant: ONE population buffer updated in place (half the memory of A/B).
//
// Let F_k(c) be the post-collision population k of cell c (what the A/B buffer holds).  The
// single buffer M stores it in one of two layouts, alternating every step:
//   after an ODD  step ("streamed"):  F_k(c) lives at M[k][c + c_k]
//   after an EVEN step ("swapped"):   F_k(c) lives at M[inv k][c]
// so an even step reads its own cell's slots and writes its own cell's slots, and an odd step
// reads the neighbours' slots and writes exactly the slots it read.  Every branch of the
// reference's boundary chain asks for some F_kk(cell) and is translated through the layout of the
// moment; bounce-back and free-slip values sit in the solid neighbour's / the ghost shell's slots,
// which no other thread touches, so the update is race-free.  The one exception is the outlet
// branch (lbm_solver.cu:292-296), which copies from another FLUID cell of the x = W-1 column and
// would race in place: those seven planes of that column are mirrored in a small double-buffered
// side array `ob` that x = W-1 cells write each step and read the step after.
// The buffer needs a ghost shell in x and y as well: Wp >= W+2 and H+2 rows per layer, addressed
// linearly (x = -1 is the last pad slot of the previous row, y = -1 the last row of the previous
// layer), which keeps x = 0 on a 128-byte boundary.
// ------------------------------------------------------------------------------------------

template <int K>
__device__ __forceinline__ long long dir_off(const Geo& g) {
    return (long long)CZ[K] * g.layer + (long long)CY[K] * g.row + CX[K];
}

// where F_KK(own cell) is stored in the layout that PHASE reads
template <int KK, int PHASE>
__device__ __forceinline__ const float* aa_own_pop(const Geo& g, const float* M, long long own) {
    return PHASE == 0 ? M + (long long)KK * g.plane + own + dir_off<KK>(g)   // streamed layout
                      : M + (long long)INV[KK] * g.plane + own;             // swapped layout
}
// where F_K(own - c_K), the plainly pulled population, is stored
template <int K, int PHASE>
__device__ __forceinline__ const float* aa_pull_pop(const Geo& g, const float* M, long long own) {
    return PHASE == 0 ? M + (long long)K * g.plane + own
                      : M + (long long)INV[K] * g.plane + own - dir_off<K>(g);
}


template <int PRESET, int K, int PHASE>
__device__ __forceinline__ float aa_gather_general(const Geo& g, const float* M, const float* __restrict__ ob_old,
                                                   const TileMask& t, int lane, int x, int y,
                                                   int zl, long long own, float inlet_e, float U) {
    const unsigned code = classify_with<PRESET, K>(g, x, y, zl, [&] { return tile_nb_solid<K>(t, lane); });
    if (code == SRC_INLET) return weight(K) * inlet_e;
    if (code == SRC_OUTLET) return aa_outlet_read<K>(g, ob_old, y, zl);
    const float* p;
    if (code == SRC_PULL) p = aa_pull_pop<K, PHASE>(g, M, own);
    else if (code == SRC_CEIL_RFY) p = aa_own_pop<RFY[K], PHASE>(g, M, own);
    else if (code == SRC_SIDE_RFZ) p = aa_own_pop<RFZ[K], PHASE>(g, M, own);
    else p = aa_own_pop<INV[K], PHASE>(g, M, own);
    float v = *p;
    if (PRESET == LBM_PRESET_CAVITY && code == SRC_LID_BB) v = v + lid_term(K, U);
    return v;
}

template <int PRESET, int K, int PHASE, int TUNE>
__device__ __forceinline__ float aa_gather_fast(const Geo& g, const float* M, const float* __restrict__ ob_old,
                                                int x, int y, int zl, long long own, float inlet_e) {
    constexpr int cx = CX[K];
    const float* p = aa_pull_pop<K, PHASE>(g, M, own);
    if (cx == 0) return ldf<TUNE>(p);
    // one load per direction, the special lane by an address select (no divergent branch); in the
    // odd phase the x-shifted loads are misaligned and keep the default cache policy like the A/B
    // fast path (ldk), in the even phase every load is an aligned own-cell slot
    if (PRESET == LBM_PRESET_WIND_TUNNEL_REF) {
        if (cx == 1) {
            const float v = PHASE ? ldk<TUNE, K>(p) : ldf<TUNE>(p);  // x = 0 reads a ghost slot: in bounds, discarded
            return x == 0 ? weight(K) * inlet_e : v;
        }
        // interior tile: ny, nz are in range
        if (x == g.W - 1) p = ob_old + ob_at(g, ob_index(K), zl - 1 - CZ[K], y - CY[K]);
    } else {
        const bool wall = (cx == 1) ? (x == 0) : (x == g.W - 1);
        if (wall) p = aa_own_pop<INV[K], PHASE>(g, M, own);
    }
    return PHASE ? ldk<TUNE, K>(p) : ldf<TUNE>(p);
}

template <int PRESET, int PHASE, int TUNE, int K>
struct AaGather {
    static __device__ __forceinline__ void fast(const Geo& g, const float* M, const float* ob, int x, int y, int zl,
                                                long long own, float e, float* f) {
        AaGather<PRESET, PHASE, TUNE, K - 1>::fast(g, M, ob, x, y, zl, own, e, f);
        f[K] = aa_gather_fast<PRESET, K, PHASE, TUNE>(g, M, ob, x, y, zl, own, e);
    }
    static __device__ __forceinline__ void general(const Geo& g, const float* M, const float* ob,
                                                   const TileMask& t, int lane, int x, int y, int zl,
                                                   long long own, float e, float U, float* f) {
        AaGather<PRESET, PHASE, TUNE, K - 1>::general(g, M, ob, t, lane, x, y, zl, own, e, U, f);
        f[K] = aa_gather_general<PRESET, K, PHASE>(g, M, ob, t, lane, x, y, zl, own, e, U);
    }
    static __device__ __forceinline__ void store(const Geo& g, float* M, long long own, const float* o) {
        AaGather<PRESET, PHASE, TUNE, K - 1>::store(g, M, own, o);
        float* dst = PHASE == 0 ? M + (long long)INV[K] * g.plane + own
                                : M + (long long)K * g.plane + own + dir_off<K>(g);
        // odd phase, x-shifted rows: two waves share the first / last 128-byte line of a segment;
        // write-back stores let L2 merge the halves, nt stores would send partial lines to HBM
        if (PHASE == 1 && CX[K] != 0) *dst = o[K];
        else stf<TUNE>(dst, o[K]);
    }
};
template <int PRESET, int PHASE, int TUNE>
struct AaGather<PRESET, PHASE, TUNE, -1> {
    static __device__ __forceinline__ void fast(const Geo&, const float*, const float*, int, int, int, long long, float, float*) {}
    static __device__ __forceinline__ void general(const Geo&, const float*, const float*, const TileMask&, int, int, int, int, long long, float, float, float*) {}
    static __device__ __forceinline__ void store(const Geo&, float*, long long, const float*) {}
};

// PHASE 0: even step (reads the streamed layout, writes the swapped one); PHASE 1: odd step.
template <int PRESET, int PHASE, int TUNE>
__global__ __launch_bounds__(256) void k_aa_step(
    float* M, const float* __restrict__ ob_old, float* __restrict__ ob_new,
    const unsigned long long* __restrict__ mask, const unsigned char* __restrict__ flags,
    float* __restrict__ rho_out, float4* __restrict__ u_out, Geo g, Params p, int segsx_log2, int zrev,
    int want_moments, int zl_begin, int zl_second) {
    const TileId t = tile_of_wave(segsx_log2, zrev);
    const int y = t.y;
    // zl_second > 0: the two slab-boundary layers (blockIdx.z = 0 -> zl_begin, 1 -> zl_second)
    const int zl = zl_second > 0 ? (t.bz == 0 ? zl_begin : zl_second)
                                 : zl_begin + ((zrev & ZORDER_REVERSE) ? (int)(gridDim.z - 1 - t.bz) : t.bz);
    if (t.xs >= g.Wt || y >= g.H) return;
    const int x = t.xs * 64 + t.lane;
    const unsigned char flag = flags[((long long)(zl - 1) * g.H + y) * g.Wt + t.xs];
    if (x >= g.W) return;
    if (flag == 2 && !want_moments) return;  // every cell of the tile is solid: nothing moves
    const long long own = cell_off(g, x, y, zl);
    const long long ridx = ((long long)(zl - 1 - g.fz0) * g.H + y) * g.W + x;
    want_moments = want_moments && (unsigned)(zl - 1 - g.fz0) < (unsigned)g.fnz;
    const float inlet_e = inlet_bracket(p.U);
    float f[Q];
    if (flag == 0) {
        AaGather<PRESET, PHASE, TUNE, Q - 1>::fast(g, M, ob_old, x, y, zl, own, inlet_e, f);
    } else {
        const TileMask tm = load_tile_mask(g, mask, t.xs, y, zl);
        if ((tm.w[1][1] >> t.lane) & 1ull) {
            if (want_moments) {
                rho_out[ridx] = 0.0f;
                u_out[ridx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            return;
        }
        AaGather<PRESET, PHASE, TUNE, Q - 1>::general(g, M, ob_old, tm, t.lane, x, y, zl, own, inlet_e, p.U, f);
    }
    const Moments m = moments_of(f);
    if (want_moments) {
        rho_out[ridx] = m.rho;
        u_out[ridx] = make_float4(m.ux, m.uy, m.uz, 0.0f);
    }
    float o[Q];
    collide_cell(f, m, p.omega_shear, p.omega_bulk, o);
    AaGather<PRESET, PHASE, TUNE, Q - 1>::store(g, M, own, o);
    if (x == g.W - 1) {
#pragma unroll
        for (int i = 0; i < OB_N; i++) ob_new[ob_at(g, i, zl - 1, y)] = o[OB_PLANE[i]];
    }
}

__global__ void k_ob_fill(float* __restrict__ a, float* __restrict__ b, long long per_plane) {
    const long long i = gidx();
    if (i >= per_plane * OB_N) return;
    const float w = weight(OB_PLANE[i / per_plane]);
    a[i] = w;
    b[i] = w;
}

// solid cells of the outlet column keep their frozen value in both side buffers
__global__ void k_ob_sync(const float* __restrict__ cur, float* __restrict__ other,
                          const unsigned long long* __restrict__ mask, Geo g) {
    const long long col = gidx();
    if (col >= (long long)g.D * g.H) return;
    const int z = (int)(col / g.H), y = (int)(col - (long long)z * g.H);
    if (!solid_at(g, mask, g.W - 1, y, z + 1)) return;
    for (int i = 0; i < OB_N; i++) other[ob_at(g, i, z, y)] = cur[ob_at(g, i, z, y)];
}

// A cell that the new mask turns from solid to fluid restarts from equilibrium (F_k = w_k): the
// in-place scheme has nowhere to keep a solid cell's frozen populations (documented deviation from
// the A/B scheme; identical to the reference whenever the cell was solid since init()).
__global__ void k_aa_refluidize(float* M, float* __restrict__ ob_cur, const unsigned long long* __restrict__ prev,
                                const unsigned long long* __restrict__ mask, Geo g, int phase) {
    const long long i = gidx();
    const long long n = (long long)g.W * g.H * g.D;
    if (i >= n) return;
    const long long row = i / g.W;
    const int x = (int)(i - row * g.W);
    const int z = (int)(row / g.H);
    const int y = (int)(row - (long long)z * g.H);
    if (!solid_at(g, prev, x, y, z + 1) || solid_at(g, mask, x, y, z + 1)) return;
    const long long own = cell_off(g, x, y, z + 1);
    for (int k = 0; k < Q; k++) {
        const long long d = (long long)CZ[k] * g.layer + (long long)CY[k] * g.row + CX[k];
        if (phase == 0) M[(long long)k * g.plane + own + d] = weight(k);
        else M[(long long)INV[k] * g.plane + own] = weight(k);
    }
    if (x == g.W - 1)
        for (int j = 0; j < OB_N; j++) ob_cur[ob_at(g, j, z, y)] = weight(OB_PLANE[j]);
}

// Sliding scheme, "pristine" mode (all solid cells hold w_k and are not carried along): write w_k into
// the slots of the cells that `mask` marks solid, at the lattice position `f` -- before somebody reads
// those slots (population read-back, halo pack) or before solid cells start to matter (mask change).
__global__ void k_slide_fill_solids(float* __restrict__ f, const unsigned long long* __restrict__ mask, Geo g) {
    const long long i = gidx();
    const long long n = (long long)g.W * g.H * g.D;
    if (i >= n) return;
    const long long row = i / g.W;
    const int x = (int)(i - row * g.W);
    const int z = (int)(row / g.H);
    const int y = (int)(row - (long long)z * g.H);
    if (!solid_at(g, mask, x, y, z + 1)) return;
    const long long own = cell_off(g, x, y, z + 1);
    for (int k = 0; k < Q; k++) f[(long long)k * g.plane + own] = weight(k);
}
// outlet column of plainly addressed populations -> both side arrays (after lbm_write_populations)
__global__ void k_ob_gather(const float* __restrict__ f, float* __restrict__ ob_a, float* __restrict__ ob_b, Geo g) {
    const long long col = gidx();
    if (col >= (long long)g.D * g.H) return;
    const int z = (int)(col / g.H), y = (int)(col - (long long)z * g.H);
    const long long own = cell_off(g, g.W - 1, y, z + 1);
    for (int j = 0; j < OB_N; j++) {
        const float v = f[(long long)OB_PLANE[j] * g.plane + own];
        ob_a[ob_at(g, j, z, y)] = v;
        ob_b[ob_at(g, j, z, y)] = v;
    }
}

// canonical f[k*n + idx] <-> the in-place layout of the moment
__global__ void k_aa_export(const float* __restrict__ M, float* __restrict__ dense, Geo g, int phase) {
    const long long i = gidx();
    const long long n = (long long)g.W * g.H * g.D;
    if (i >= n) return;
    const long long row = i / g.W;
    const int x = (int)(i - row * g.W);
    const int z = (int)(row / g.H);
    const int y = (int)(row - (long long)z * g.H);
    const long long own = cell_off(g, x, y, z + 1);
    for (int k = 0; k < Q; k++) {
        const long long d = (long long)CZ[k] * g.layer + (long long)CY[k] * g.row + CX[k];
        dense[(long long)k * n + i] = phase == 0 ? M[(long long)k * g.plane + own + d] : M[(long long)INV[k] * g.plane + own];
    }
}
__global__ void k_aa_import(float* __restrict__ M, float* __restrict__ ob_a, float* __restrict__ ob_b,
                            const float* __restrict__ dense, Geo g) {
    const long long i = gidx();
    const long long n = (long long)g.W * g.H * g.D;
    if (i >= n) return;
    const long long row = i / g.W;
    const int x = (int)(i - row * g.W);
    const int z = (int)(row / g.H);
    const int y = (int)(row - (long long)z * g.H);
    const long long own = cell_off(g, x, y, z + 1);
    for (int k = 0; k < Q; k++) {
        const long long d = (long long)CZ[k] * g.layer + (long long)CY[k] * g.row + CX[k];
        M[(long long)k * g.plane + own + d] = dense[(long long)k * n + i];  // streamed layout
    }
    if (x == g.W - 1)
        for (int j = 0; j < OB_N; j++) {
            const float v = dense[(long long)OB_PLANE[j] * n + i];
            ob_a[ob_at(g, j, z, y)] = v;
            ob_b[ob_at(g, j, z, y)] = v;
        }
}
