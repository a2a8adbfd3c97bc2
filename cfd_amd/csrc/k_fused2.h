// k_fused2.h -- TWO time steps per pass over HBM (in-kernel temporal blocking) for the A/B scheme.
// Part of the single translation unit lbm_hip.hip (included inside its anonymous namespace, after
// k_step.h); not a stand-alone header.
//
// Why: the one-step kernel runs at the rate this part sustains for equal bytes in and out (6.2-6.3 TB/s,
// 0.78 of the 8 TB/s spec).  More updates per second need fewer bytes per update:
// here a cell's populations are read once and written once per TWO updates (plus the tile halo), i.e.
// 76 B x (1 + halo) + 76 B instead of 304 B.
//
// How: a workgroup of TY + 3 waves owns a 64 x TY tile of (x, y) and marches through a chunk of z layers.
// TY + 2 "row" waves hold one row each of the tile plus one halo row below and above; one more wave holds
// the two halo columns (x0 - 1 and x0 + 64) of those rows.  Per layer L every wave computes the FIRST step
// for its cells -- the same pull gather + collision as k_stream_collide's interior path -- and keeps the
// result in registers for three layers.  What the neighbours need for the SECOND step of layer L - 1 (the
// up-going oblique populations of layer L - 2, the in-plane ones of layer L - 1, the down-going oblique ones
// of layer L: 16 values per cell) goes through LDS; the own-column ones (rest, +z, -z) never leave the
// registers.  The TY inner row waves then gather the second step from LDS, collide and store time t + 2.
// The next layer's 19 loads travel global -> LDS (global_load_lds_dword, no registers held in flight)
// while all this happens.
//
// Scope: tiles whose haloed neighbourhood is plain fluid (tile flag 0 everywhere: no wall row, no solid cell
// in reach), with the x ends of the domain handled in place exactly as k_stream_collide's interior path
// does (inlet constant / outlet copy of the reference chain, lbm_solver.cu:283-296, or the cavity's wall
// bounce-back).  Every other cell -- the rim -- takes two ordinary steps through a third buffer (host side:
// fused_pair() in lbm_hip.hip).  The arithmetic per cell is moments_of() + collide_cell() on the same
// operands in the same order as two single steps, so results are bit-identical (tests/test_fused_gpu.py).

// LBM_F2_SCALAR_COLLIDE: the scalar moments_of() + collide_cell() instead of collide_packed() (same bits)
#ifdef LBM_F2_SCALAR_COLLIDE
#define F2_COLLIDE(IN, OUT) { const Moments m_ = moments_of(IN); collide_cell(IN, m_, p.omega_shear, p.omega_bulk, OUT); }
#else
#define F2_COLLIDE(IN, OUT) collide_packed(IN, p.omega_shear, p.omega_bulk, OUT)
#endif

struct FusedGrid {
    int nbx, nby, nbz;      // 64-cell tiles in x, TY-row tiles in y, pieces an unobstructed tile column is cut into in z
    int nblocks;            // blocks = workgroups of a launch
    int y_first, y_end;     // rows [y_first, y_end) take the fused path (y_first >= 2, y_end <= H - 2)
    int zl_first, zl_end;   // local layers (ghost offset included) the blocks lie in: [3, D - 1)
    int zchunk;             // most layers a block takes
};

constexpr int F2_LCOLS = 66;   // 64 cells + one halo column each side
#ifndef F2_XCD_REMAP
#define F2_XCD_REMAP 1
#endif
// LDS plane of an exchanged population: 0..3 up-going oblique (11,14,15,18), 4..11 in-plane (1,2,3,4,7,8,9,10),
// 12..15 down-going oblique (12,13,16,17); -1: own column (0, 5, 6), stays in registers
__host__ __device__ constexpr int f2_plane(int k) {
    return k == 11 ? 0 : k == 14 ? 1 : k == 15 ? 2 : k == 18 ? 3 :
           k == 1 ? 4 : k == 2 ? 5 : k == 3 ? 6 : k == 4 ? 7 : k == 7 ? 8 : k == 8 ? 9 : k == 9 ? 10 : k == 10 ? 11 :
           k == 12 ? 12 : k == 13 ? 13 : k == 16 ? 14 : k == 17 ? 15 : -1;
}
__host__ __device__ constexpr size_t f2_lds_bytes(int TY, int STAGE, int NBUF) {
    return (size_t)(NBUF * 16 * (TY + 2) * F2_LCOLS + (STAGE ? (TY + 3) * Q * 64 : 0)) * sizeof(float);
}

// Which wave plays which part.  Waves go to the four SIMDs of a CU round-robin, and a SIMD's waves share
// its issue slots: the three waves that only do the first step (the two halo rows and the halo columns)
// are placed so that no SIMD carries more collisions than it must.
template <int TY>
__device__ __forceinline__ int f2_role(int wave) {   // 0 .. TY+1: row of the haloed tile; TY+2: halo columns
    constexpr int NW = TY + 3;
    constexpr int stride = (NW % 4 == 1) ? 4 : 1;     // NW = 4n+1: SIMD 0 has one wave more -- give it the light ones
    if (wave == 0) return 0;
    if (wave == stride) return TY + 1;
    if (wave == 2 * stride) return TY + 2;
    int before = 1 + (wave > stride) + (wave > 2 * stride);
    return wave - before + 1;
}

// first step, one direction: k_stream_collide's interior gather (gather_fast) as base + 32-bit offset.
// `base` = start of the chunk's lowest input layer; `off` / `offm` = byte offset of the lane's cell from it
// (offm: one float to the left for the outlet lane x = W-1, whose cx = -1 directions copy from x = W-1).
template <int PRESET, int K>
__device__ __forceinline__ const float* f2_src(const char* __restrict__ base, const Geo& g, unsigned off, unsigned offm,
                                               bool xl, bool xr) {
    constexpr int cx = CX[K];
    const long long dk = ((long long)K * g.plane - (long long)CZ[K] * g.layer - (long long)CY[K] * g.row - cx) * 4;
    if (cx == 0) return reinterpret_cast<const float*>(base + dk + off);
    if (PRESET == LBM_PRESET_WIND_TUNNEL_REF)
        return reinterpret_cast<const float*>(base + dk + (cx == -1 ? offm : off));
    // cavity: the wall lane reads its own cell's opposite population
    const long long dw = (long long)INV[K] * g.plane * 4;
    const bool wall = cx == 1 ? xl : xr;
    const unsigned delta = (unsigned)(int)(dw - dk);          // exact in 32 bits: |dw - dk| < 2 layers (checked on the host)
    return reinterpret_cast<const float*>(base + dk + (wall ? off + delta : off));
}

template <int PRESET, int K>
struct F2 {
    // issue the first-step loads of one layer into the wave's staging area (LDS-DMA)
    static __device__ __forceinline__ void issue(const char* __restrict__ base, const Geo& g, unsigned off, unsigned offm,
                                                 bool xl, bool xr, float* stage) {
        F2<PRESET, K - 1>::issue(base, g, off, offm, xl, xr, stage);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)f2_src<PRESET, K>(base, g, off, offm, xl, xr),
                                         (__attribute__((address_space(3))) void*)(stage + K * 64), 4, 0, 0);
    }
    // the same loads straight into registers (no prefetch)
    static __device__ __forceinline__ void load(const char* __restrict__ base, const Geo& g, unsigned off, unsigned offm,
                                                bool xl, bool xr, float* f) {
        F2<PRESET, K - 1>::load(base, g, off, offm, xl, xr, f);
        f[K] = ldk<1, K>(f2_src<PRESET, K>(base, g, off, offm, xl, xr));
    }
    // staged values -> registers; the inlet lane's cx = +1 directions are the constant inlet equilibrium
    static __device__ __forceinline__ void unstage(const float* stage, int lane, bool xl, float inlet_e, float* f) {
        F2<PRESET, K - 1>::unstage(stage, lane, xl, inlet_e, f);
        f[K] = stage[K * 64 + lane];
    }
    static __device__ __forceinline__ void fix_inlet(bool xl, float inlet_e, float* f) {
        F2<PRESET, K - 1>::fix_inlet(xl, inlet_e, f);
        if (PRESET == LBM_PRESET_WIND_TUNNEL_REF && CX[K] == 1) f[K] = xl ? weight(K) * inlet_e : f[K];
    }
    // what the neighbours need of this cell for the second step of layer z: up-going values of layer z-1,
    // in-plane values of layer z, down-going values of layer z+1
    template <int PLANE>
    static __device__ __forceinline__ void put(float* exch, int slot, const float* below, const float* cur, const float* above) {
        F2<PRESET, K - 1>::template put<PLANE>(exch, slot, below, cur, above);
        constexpr int pl = f2_plane(K);
        if (pl >= 0) exch[pl * PLANE + slot] = CZ[K] == 1 ? below[K] : (CZ[K] == 0 ? cur[K] : above[K]);
    }
    // second step, one direction: the streamed population of cell (col, r) of the tile, with the x ends of
    // the domain handled as in the first step (slotm: the outlet lane reads its own column)
    template <int PLANE>
    static __device__ __forceinline__ void get(const float* exch, int slot, int slotm, bool xl, bool xr, float inlet_e,
                                               const float* below, const float* cur, const float* above, float* f) {
        F2<PRESET, K - 1>::template get<PLANE>(exch, slot, slotm, xl, xr, inlet_e, below, cur, above, f);
        constexpr int pl = f2_plane(K);
        constexpr int cx = CX[K];
        if (pl < 0) {
            f[K] = K == 0 ? cur[0] : (K == 5 ? below[5] : above[6]);
        } else if (PRESET == LBM_PRESET_WIND_TUNNEL_REF) {
            const float v = exch[pl * PLANE - CY[K] * F2_LCOLS - cx + (cx == -1 ? slotm : slot)];
            f[K] = (cx == 1 && xl) ? weight(K) * inlet_e : v;
        } else {
            const float v = exch[pl * PLANE - CY[K] * F2_LCOLS - cx + slot];
            const bool wall = cx == 1 ? xl : (cx == -1 ? xr : false);
            f[K] = wall ? cur[INV[K]] : v;    // wall bounce-back: the own cell's opposite population of the same layer
        }
    }
};
template <int PRESET>
struct F2<PRESET, -1> {
    static __device__ __forceinline__ void issue(const char*, const Geo&, unsigned, unsigned, bool, bool, float*) {}
    static __device__ __forceinline__ void load(const char*, const Geo&, unsigned, unsigned, bool, bool, float*) {}
    static __device__ __forceinline__ void unstage(const float*, int, bool, float, float*) {}
    static __device__ __forceinline__ void fix_inlet(bool, float, float*) {}
    template <int PLANE> static __device__ __forceinline__ void put(float*, int, const float*, const float*, const float*) {}
    template <int PLANE> static __device__ __forceinline__ void get(const float*, int, int, bool, bool, float, const float*,
                                                                    const float*, const float*, float*) {}
};

// TY: rows per tile.  STAGE 1: the next layer's loads are prefetched into LDS; 0: loaded when needed.
// NBUF 2: two exchange buffers, one barrier per layer; 1: one buffer, two barriers per layer.
template <int PRESET, int TY, int STAGE, int NBUF>
__global__ __launch_bounds__((TY + 3) * 64) void k_fused2(const float* __restrict__ fin, float* __restrict__ fout,
                                                          const unsigned long long* __restrict__ blocks, Geo g, Params p,
                                                          FusedGrid fg) {
    constexpr int ROWS = TY + 2;
    constexpr int PLANE = ROWS * F2_LCOLS;
    constexpr int EXCH = 16 * PLANE;
    extern __shared__ float lds[];
    // One workgroup per block of the host's list (fused_prepare() in lbm_hip.hip: a 64 x TY tile and a run of
    // layers whose haloed neighbourhood is plain fluid, ordered layer range / y tile / x tile; the rim path owns
    // every other cell).  The hardware deals workgroups to the 8 XCDs round-robin in launch
    // order, and each XCD has its own L2.  Neighbouring tiles read each other's edge lines (the halo rows and
    // columns, the second 128-byte line of an x-shifted load), so every XCD takes ONE contiguous range of the
    // list: the lines two tiles share are then fetched from HBM once instead of once per L2, and every XCD
    // still gets the same number of blocks whatever the obstacles leave eligible.
    int bx, by, zs, ze;                                          // second step: local layers [zs, ze)
    {
        const unsigned total = gridDim.x, launch = blockIdx.x;
        const unsigned xcd = launch & 7u, nth = launch >> 3, q = total >> 3, r = total & 7u;
        const unsigned long long e = blocks[F2_XCD_REMAP ? xcd * q + min(xcd, r) + nth : launch];
        // the entry is the same for the whole workgroup: keep what is derived from it in scalar registers
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)e), hi = __builtin_amdgcn_readfirstlane((unsigned)(e >> 32));
        bx = (int)(lo & 0xfffu);
        by = (int)((lo >> 12) & 0xfffu);
        zs = (int)((lo >> 24) | ((hi & 0xfffu) << 8));
        // (the host never lets a block run past zl_end; written as a min() the compiler cannot derive a minimum trip
        // count of the layer loop from the entry's layout, which made it peel the loop and spill 18 registers)
        ze = min(zs + (int)(hi >> 12), fg.zl_end);
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int role = f2_role<TY>(wave);
    const int x0 = bx * 64, y0 = fg.y_first + by * TY;
    const int tyn = min(TY, fg.y_end - y0);                     // rows of this tile that exist (the last tile may be short)

    int x, r;
    bool active;
    if (role < ROWS) { r = role; x = x0 + lane; active = true; }
    else { r = lane < ROWS ? lane : lane - ROWS; x = lane < ROWS ? x0 - 1 : x0 + 64; active = lane < 2 * ROWS; }
    active = active && r <= tyn + 1 && x >= 0 && x < g.W;
    const bool second = role >= 1 && role <= tyn && x < g.W;
    const int y = y0 - 1 + r;
    const int col = x - x0 + 1;
    const int slot = r * F2_LCOLS + col;
    const bool xl = x == 0, xr = x == g.W - 1;
    const int slotm = slot - (xr ? 1 : 0);                       // outlet lane: its cx = -1 directions copy from its own column
    const float inlet_e = inlet_bracket(p.U);
    float* stage = lds + NBUF * EXCH + wave * (Q * 64);

    // input addressing: byte offsets from layer zs - 3 (two layers of slack keep every offset, the wall and
    // outlet adjustments included, non-negative)
    const char* base_in = reinterpret_cast<const char*>(fin + (long long)(zs - 3) * g.layer);
    char* base_out = reinterpret_cast<char*>(fout + (long long)(zs - 3) * g.layer);
    const unsigned layer_b = (unsigned)(g.layer * 4), plane_b = (unsigned)(g.plane * 4);
    const unsigned cell_b = (unsigned)(((long long)(active ? y : y0) * g.row + (active ? x : x0)) * 4);
    auto off_of = [&](int L) { return (unsigned)(L - (zs - 3)) * layer_b + cell_b; };

    float below[Q], cur[Q], above[Q], pf[Q];   // first-step results of layers L-2, L-1, L; the loads of layer L
#pragma unroll
    for (int k = 0; k < Q; k++) { below[k] = 0.f; cur[k] = 0.f; above[k] = 0.f; pf[k] = 0.f; }
    if (STAGE) {
        const unsigned off = off_of(zs - 1);
        if (active) F2<PRESET, Q - 1>::issue(base_in, g, off, off - (xr ? 4u : 0u), xl, xr, stage);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    int buf = 0;
    for (int L = zs - 1; L <= ze; L++) {
        // ---- first step of layer L ---------------------------------------------------------------------
        if (active) {
            if (STAGE) {
                F2<PRESET, Q - 1>::unstage(stage, lane, xl, inlet_e, pf);
            } else {
                const unsigned off = off_of(L);
                F2<PRESET, Q - 1>::load(base_in, g, off, off - (xr ? 4u : 0u), xl, xr, pf);
            }
            F2<PRESET, Q - 1>::fix_inlet(xl, inlet_e, pf);
            F2_COLLIDE(pf, above);
        }
        if (STAGE) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging area has been read: refill it
            if (active && L < ze) {
                const unsigned off = off_of(L + 1);
                F2<PRESET, Q - 1>::issue(base_in, g, off, off - (xr ? 4u : 0u), xl, xr, stage);
            }
        }
        // ---- second step of layer L - 1 ------------------------------------------------------------------
        if (L >= zs + 1) {
            float* exch = lds + buf * EXCH;
            if (active) F2<PRESET, Q - 1>::template put<PLANE>(exch, slot, below, cur, above);
            __syncthreads();
            if (second) {
                float f[Q], o[Q];
                F2<PRESET, Q - 1>::template get<PLANE>(exch, slot, slotm, xl, xr, inlet_e, below, cur, above, f);
                F2_COLLIDE(f, o);
                // the staged loads have had the whole second step to land; waiting for them HERE keeps the
                // latency of the stores below out of the next iteration (stores count in vmcnt too)
                if (STAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned offo = off_of(L - 1);
#pragma unroll
                for (int k = 0; k < Q; k++)   // one base, 32-bit offsets: population k sits k planes further on
                    __builtin_nontemporal_store(o[k], reinterpret_cast<float*>(base_out + (offo + (unsigned)k * plane_b)));
            } else {
                if (STAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (NBUF == 1) __syncthreads(); else buf ^= 1;
        } else {
            if (STAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int k = 0; k < Q; k++) { below[k] = cur[k]; cur[k] = above[k]; }
    }
}
