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
// Scope: blocks of tiles whose haloed neighbourhood keeps off the domain's walls (no wall row, no wall layer; with the
// optional WALL form of the kernel the blocks may lie on them, see F2Wall),
// with the x ends of the domain handled in place exactly as k_stream_collide's interior path does (inlet
// constant / outlet copy of the reference chain, lbm_solver.cu:283-296, or the cavity's wall bounce-back), and
// -- in the OBST form of the kernel -- with solid cells inside: a direction that would pull from a solid cell
// takes the own cell's opposite population (lbm_solver.cu:298-318).  Every other cell -- the rim -- takes two
// ordinary steps through a third buffer (host side: fused_pair() in lbm_hip.hip).  The arithmetic per cell
// is collide_packed() (d3q19.h): moments_of() + collide_cell() on the same operands in the same order, so
// results are bit-identical to two single steps (tests/test_fused_gpu.py).

// LBM_F2_SCALAR_COLLIDE: the scalar moments_of() + collide_cell() instead of collide_packed() (same bits)
#ifdef LBM_F2_SCALAR_COLLIDE
#define F2_COLLIDE(IN, OUT) { const Moments m_ = moments_of(IN); collide_cell(IN, m_, p.omega_shear, p.omega_bulk, OUT); }
#else
#define F2_COLLIDE(IN, OUT) collide_packed(IN, p.omega_shear, p.omega_bulk, OUT)
#endif

struct FusedGrid {
    int nbx, nby;           // 64-cell tiles in x, tiles of at most TY rows in y (rows dealt evenly: fused_prepare())
    int nblocks;            // workgroups of a launch: 8 x the longest of the eight XCD ranges of the block list
    int y_first, y_end;     // rows [y_first, y_end) take the fused path (y_first >= 2, y_end <= H - 2)
    int zl_first, zl_end;   // local layers (ghost offset included) the blocks lie in: [3, D - 1)
    int zchunk;             // most layers a block takes
    int wide;               // plain blocks take the wide form of the first-step loads (LBM_FUSED_WIDE=0 turns it off)
};

// One of the rim's two steps can ride in the fused launch: it writes cells no block touches, and what it reads is
// complete before the launch starts.  Undecomposed contexts: the FIRST step (the rim tiles and the tiles next to them,
// t -> t+1 from the buffer the blocks read into the intermediate one).  z-slabs: the SECOND (t+1 -> t+2 from the
// intermediate buffer into the blocks' output buffer; the first must end early there, the boundary layers' second step
// waits for it).  The workgroups BEHIND the last block of the list take that step's tile list, one entry per wave
// (k_stream_collide's MODE 4 entries: xs | y << 16 | zl << 40, ~0 pads).  They are dispatched when the blocks have all
// started, i.e. onto the CUs that hold no block and onto those whose block has ended -- the ragged tail of a launch of
// several rounds -- and one kernel boundary of a pair of steps goes away (profiles/r03_tuning.md section 13).
struct F2Rim {
    const unsigned long long* list;   // nullptr / n = 0: both steps of the rim have launches of their own
    const unsigned long long* mask;
    const unsigned char* flags;
    const float* src;
    float* dst;
    unsigned n;
};
template <int PRESET>
__device__ __forceinline__ void f2_rim_tile(const F2Rim& rim, const Geo& g, const Params& p, unsigned long long e, int lane) {
    const float* __restrict__ fin = rim.src;
    if (e == ~0ull) return;
    const int xs = (int)(e & 0xffffu), y = (int)((e >> 16) & 0xffffffu), zl = (int)(e >> 40);
    if (xs >= g.Wt || y >= g.H) return;
    const int x = xs * 64 + lane;
    const unsigned char flag = rim.flags[((long long)(zl - 1) * g.H + y) * g.Wt + xs];
    if (x >= g.W || flag == 2) return;
    const long long own = cell_off(g, x, y, zl);
    const float inlet_e = inlet_bracket(p.U);
    float f[Q];
    if (flag == 0) {
        GatherAll<PRESET, Q - 1>::template fast<1>(g, fin, x, y, zl, own, inlet_e, f);
    } else {
        const TileMask tm = load_tile_mask(g, rim.mask, xs, y, zl);
        if ((tm.w[1][1] >> lane) & 1ull) return;      // a solid cell: frozen in every buffer (k_sync_solid)
        GatherAll<PRESET, Q - 1>::template general<false>(g, fin, tm, lane, x, y, zl, own, inlet_e, p.U, nullptr, f);
    }
    const Moments m = moments_of(f);
    float o[Q];
    collide_cell(f, m, p.omega_shear, p.omega_bulk, o);
    float* dst = rim.dst + own;
#pragma unroll
    for (int k = 0; k < Q; k++) stf<1>(dst + (long long)k * g.plane, o[k]);
}

// populations whose first-step loads carry the non-temporal hint (bit k), as LBM_NT_SET for the one-step kernel
#ifndef F2_NT_SET
#define F2_NT_SET 0u
#endif
constexpr int F2_LCOLS = 66;   // 64 cells + one halo column each side
// LDS plane of an exchanged population: 0..3 up-going oblique (11,14,15,18), 4..11 in-plane (1,2,3,4,7,8,9,10),
// 12..15 down-going oblique (12,13,16,17); -1: own column (0, 5, 6), stays in registers
__host__ __device__ constexpr int f2_plane(int k) {
    return k == 11 ? 0 : k == 14 ? 1 : k == 15 ? 2 : k == 18 ? 3 :
           k == 1 ? 4 : k == 2 ? 5 : k == 3 ? 6 : k == 4 ? 7 : k == 7 ? 8 : k == 8 ? 9 : k == 9 ? 10 : k == 10 ? 11 :
           k == 12 ? 12 : k == 13 ? 13 : k == 16 ? 14 : k == 17 ? 15 : -1;
}
// staging area of a wave: 64 floats per population, or F2_WIDE = 68 in the wide form of the loads (WIDE below)
constexpr int F2_WIDE = 68;
__host__ __device__ constexpr bool f2_wide_ok(int STAGE, int NBUF) { return STAGE == 1 && NBUF == 1; }
__host__ __device__ constexpr size_t f2_lds_bytes(int TY, int STAGE, int NBUF) {
    return (size_t)(NBUF * 16 * (TY + 2) * F2_LCOLS + (STAGE ? (TY + 3) * Q * (f2_wide_ok(STAGE, NBUF) ? F2_WIDE : 64) : 0)) * sizeof(float);
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

template <int TY>
__device__ __forceinline__ int f2_wave_of(int role) {   // inverse of f2_role
    int w = 0;
    for (int i = 0; i < TY + 3; i++) w = f2_role<TY>(i) == role ? i : w;
    return w;
}

// first step, one direction: k_stream_collide's interior gather (gather_fast) as base + 32-bit offset.
// `base` = start of the chunk's lowest input layer; `off` / `offm` = byte offset of the lane's cell from it
// (offm: one float to the left for the outlet lane x = W-1, whose cx = -1 directions copy from x = W-1).
// The x shift (-cx floats) is NOT part of the address: the caller applies it as the instruction's immediate offset
// (F2_XOFF<K> bytes), so that the scalar base + 32-bit lane offset form of the load survives (the compiler otherwise
// builds 64-bit lane addresses for the ten x-shifted directions: two v_lshl_add_u64 each per layer).
#ifndef F2_WIDE_LOADS
#define F2_WIDE_LOADS 1   // 0: the wide form of the loads is not compiled in
#endif
#ifndef F2_OPT
#define F2_OPT 8    // measured at 256^3 and 512x256x256 + car (profiles/r03_tuning.md): only bit 3 is worth having. bit 0: x shift as immediate offset, 1: unconditional exchange reads, 2: scalar store bases, 3: per-iteration M0 / plane stride
#endif
template <int K>
constexpr int F2_XOFF = (F2_OPT & 1) ? -CX[K] * 4 : 0;
// Walls (WALL form of the kernel): which of the domain's y / z walls the lane's cell of the current layer lies on.
// A direction that would pull across one takes a population of the own cell instead, like a solid neighbour's
// bounce-back, and the solid-neighbour word carries its bit; only the plane differs (lbm_solver.cu:298-312: floor
// bounce-back, ceiling reflect_y, side reflect_z, tested in that order before any solid; the cavity bounces back
// from every wall and its lid adds the lid term).
struct F2Wall {
    bool bot, top, zlo, zhi;
};
constexpr unsigned f2_dirs(int cy_sel, int cz_sel) {   // bit k: CY[k] == cy_sel (cz_sel == 2) or CZ[k] == cz_sel (cy_sel == 2)
    unsigned m = 0;
    for (int k = 0; k < Q; k++)
        if ((cz_sel == 2 && CY[k] == cy_sel) || (cy_sel == 2 && CZ[k] == cz_sel)) m |= 1u << k;
    return m;
}
constexpr unsigned F2_CY_POS = f2_dirs(1, 2), F2_CY_NEG = f2_dirs(-1, 2), F2_CZ_POS = f2_dirs(2, 1), F2_CZ_NEG = f2_dirs(2, -1);
// the own cell's population that direction K takes when it does not pull (0: INV, 1: RFY, 2: RFZ)
template <int PRESET, int K>
__device__ __forceinline__ int f2_own_kind(const F2Wall& w) {
    if (PRESET != LBM_PRESET_WIND_TUNNEL_REF) return 0;
    if (CY[K] == 1 && w.bot) return 0;
    if (CY[K] == -1 && w.top) return 1;
    if ((CZ[K] == 1 && w.zlo) || (CZ[K] == -1 && w.zhi)) return 2;
    return 0;
}

template <int PRESET, int K, int WALL>
__device__ __forceinline__ const float* f2_src(const char* __restrict__ base, const Geo& g, unsigned off, unsigned offm,
                                               bool xl, bool xr, bool bb, const F2Wall& w) {
    constexpr int cx = CX[K];
    const long long dk = ((long long)K * g.plane - (long long)CZ[K] * g.layer - (long long)CY[K] * g.row) * 4 - (cx * 4 + F2_XOFF<K>);
    // bounce-back (a cavity wall, or with OBST a solid neighbour): the own cell's opposite population, no x shift
    const long long dw = (long long)INV[K] * g.plane * 4 - F2_XOFF<K>;
    unsigned delta = (unsigned)(int)(dw - dk);                    // exact in 32 bits: |dw - dk| < 2 layers (checked on the host)
    if (WALL && PRESET == LBM_PRESET_WIND_TUNNEL_REF && (CY[K] == -1 || CZ[K] != 0)) {
        const int kind = f2_own_kind<PRESET, K>(w);
        const unsigned dy = (unsigned)(int)((long long)RFY[K] * g.plane * 4 - F2_XOFF<K> - dk);
        const unsigned dz = (unsigned)(int)((long long)RFZ[K] * g.plane * 4 - F2_XOFF<K> - dk);
        delta = kind == 1 ? dy : (kind == 2 ? dz : delta);
    }
    if (cx == 0) return reinterpret_cast<const float*>(base + dk + (bb ? off + delta : off));
    if (PRESET == LBM_PRESET_WIND_TUNNEL_REF)
        return reinterpret_cast<const float*>(base + dk + (bb ? off + delta : (cx == -1 ? offm : off)));
    const bool wall = cx == 1 ? xl : xr;
    return reinterpret_cast<const float*>(base + dk + ((wall || bb) ? off + delta : off));
}

// OBST: which directions of a cell bounce back from a solid neighbour (bit k; lbm_solver.cu:298-318 via
// classify_with() in k_step.h; bit 0: the cell itself is solid) -- one 32-bit word per cell, computed once per
// mask change by k_f2_nb_words into an array laid out like one population of the planes layout ([zl][y][x],
// row pitch Wp, ghost layers included).  The fused kernel reads the word of a cell two layers ahead of its
// use (4 B per cell and pass beside the 152 B of the populations), so that no wave ever waits for the mask:
// a first version that walked the packed mask's nine rows per flagged tile inside the layer loop stalled its
// whole workgroup at the barrier for the latency of those loads (36 GLUPS inside the kernel instead of 57).
template <int K>
struct F2Nb {
    static __device__ __forceinline__ unsigned cell(const Geo& g, const unsigned long long* __restrict__ mask, int x, int y, int zl) {
        const int nx = x - CX[K], ny = y - CY[K];
        unsigned m = F2Nb<K - 1>::cell(g, mask, x, y, zl);
        if (nx >= 0 && nx < g.W && ny >= 0 && ny < g.H) m |= (unsigned)solid_at(g, mask, nx, ny, zl - CZ[K]) << K;
        return m;
    }
};
template <>
struct F2Nb<0> {
    static __device__ __forceinline__ unsigned cell(const Geo& g, const unsigned long long* __restrict__ mask, int x, int y, int zl) {
        return (unsigned)solid_at(g, mask, x, y, zl);
    }
};
__global__ __launch_bounds__(256) void k_f2_nb_words(const unsigned long long* __restrict__ mask, unsigned* __restrict__ nbw, Geo g) {
    const long long i = gidx();                       // over the owned layers zl = 1 .. D
    const long long n = (long long)g.D * g.H * g.Wp;
    if (i >= n) return;
    const int x = (int)(i % g.Wp);
    const int y = (int)((i / g.Wp) % g.H);
    const int zl = (int)(i / ((long long)g.Wp * g.H)) + 1;
    nbw[(long long)g.H * g.Wp + i] = x < g.W ? F2Nb<Q - 1>::cell(g, mask, x, y, zl) : 0u;
}

// direction K of a cell bounces back from a solid neighbour; the outlet branch (x = W-1, cx = -1) comes before
// the solid test in the reference chain
template <int K>
__device__ __forceinline__ bool f2_bb(unsigned nb, bool xr) {
    return ((nb >> K) & 1u) && !(CX[K] == -1 && xr);
}

template <int PRESET, int K>
struct F2 {
    // issue the first-step loads of one layer into the wave's staging area (LDS-DMA)
    // (`stage`: LDS byte address of the wave's staging area)
    template <int WALL>
    static __device__ __forceinline__ void issue(const char* __restrict__ base, const Geo& g, unsigned off, unsigned offm,
                                                 bool xl, bool xr, unsigned nb, unsigned stage, const F2Wall& w) {
        F2<PRESET, K - 1>::template issue<WALL>(base, g, off, offm, xl, xr, nb, stage, w);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)f2_src<PRESET, K, WALL>(base, g, off, offm, xl, xr, f2_bb<K>(nb, xr), w),
                                         // (the instruction's immediate offset moves the LDS address as well: taken off M0 again)
                                         (__attribute__((address_space(3))) void*)(size_t)(stage + K * 256u - F2_XOFF<K>), 4, F2_XOFF<K>, ((F2_NT_SET >> K) & 1u) ? 2 : 0);
    }
    // WIDE form (plain blocks): lanes 0..16 of a row wave load 16 bytes each = the 68 floats x0-2 .. x0+65 of the source
    // row of direction K, x shift left to the unstage; `base` + `offw` = lane's 16 bytes of the row of cell (x0, y, L).
    // The halo-column wave then issues no loads at all: it takes its cells' values from the row waves' areas.
    static __device__ __forceinline__ void issue_wide(const char* __restrict__ base, const Geo& g, unsigned offw, unsigned stage) {
        F2<PRESET, K - 1>::issue_wide(base, g, offw, stage);
        // (the 68-float window starts at x0-4 for cx = +1, x0-2 for cx = 0, x0 for cx = -1: what the 66 cells of the row
        // pull from then lies in three lines instead of four for the x-shifted directions, and on 16-byte boundaries)
        const long long dk = ((long long)K * g.plane - (long long)CZ[K] * g.layer - (long long)CY[K] * g.row) * 4 - CX[K] * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + dk + offw),
                                         (__attribute__((address_space(3))) void*)(size_t)(stage + K * (F2_WIDE * 4u)), 16, 0, ((F2_NT_SET >> K) & 1u) ? 2 : 0);
    }
    // `src`: the staging area of the wave that holds the lane's row, at the lane's column (x - (x0 - 2)); `adj`: 1 for the
    // outlet lane, whose cx = -1 directions copy from its own column
    static __device__ __forceinline__ void unstage_wide(const float* src, int adj, float* f) {
        F2<PRESET, K - 1>::unstage_wide(src, adj, f);
        // cell at column c = x - (x0 - 2) pulls from float x - cx, which sits at slot c - cx + 2 cx of its direction's window
        if (PRESET == LBM_PRESET_WIND_TUNNEL_REF && CX[K] == -1) f[K] = src[K * F2_WIDE - 1 - adj];
        else f[K] = src[K * F2_WIDE + CX[K]];
    }
    // the same loads straight into registers (no prefetch)
    template <int WALL>
    static __device__ __forceinline__ void load(const char* __restrict__ base, const Geo& g, unsigned off, unsigned offm,
                                                bool xl, bool xr, unsigned nb, float* f, const F2Wall& w) {
        F2<PRESET, K - 1>::template load<WALL>(base, g, off, offm, xl, xr, nb, f, w);
        f[K] = ldk<1, K>(f2_src<PRESET, K, WALL>(base, g, off, offm, xl, xr, f2_bb<K>(nb, xr), w) + F2_XOFF<K> / 4);
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
    // the cavity's lid row: the bounced-back population gains the lid term (lbm_solver.cu's moving-wall branch as
    // k_stream_collide's SRC_LID_BB has it: v + lid_term, for every direction that points down from the lid)
    static __device__ __forceinline__ void fix_lid(bool top, float U, float* f) {
        F2<PRESET, K - 1>::fix_lid(top, U, f);
        if (PRESET == LBM_PRESET_CAVITY && CY[K] == -1) f[K] = top ? f[K] + lid_term(K, U) : f[K];
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
    template <int PLANE, int WALL>
    static __device__ __forceinline__ void get(const float* exch, int slot, int slotm, bool xl, bool xr, float inlet_e, unsigned nb,
                                               const float* below, const float* cur, const float* above, float* f, const F2Wall& w, float U) {
        F2<PRESET, K - 1>::template get<PLANE, WALL>(exch, slot, slotm, xl, xr, inlet_e, nb, below, cur, above, f, w, U);
        constexpr int pl = f2_plane(K);
        constexpr int cx = CX[K];
        const bool bb = f2_bb<K>(nb, xr);     // solid neighbour (or a wall): a population of the own cell, same layer
        float own = cur[INV[K]];
        if (WALL && PRESET == LBM_PRESET_WIND_TUNNEL_REF && (CY[K] == -1 || CZ[K] != 0)) {
            const int kind = f2_own_kind<PRESET, K>(w);
            own = kind == 1 ? cur[RFY[K]] : (kind == 2 ? cur[RFZ[K]] : own);
        }
        if (WALL && PRESET == LBM_PRESET_CAVITY && CY[K] == -1) own = w.top ? own + lid_term(K, U) : own;
        if (pl < 0) {
            const float v = K == 0 ? cur[0] : (K == 5 ? below[5] : above[6]);
            f[K] = (K != 0 && bb) ? own : v;
        } else if (PRESET == LBM_PRESET_WIND_TUNNEL_REF) {
            float v = exch[pl * PLANE - CY[K] * F2_LCOLS - cx + (cx == -1 ? slotm : slot)];
            // (read by every lane: left to itself the compiler masks the inlet lane out of the read, which costs
            // three scalar instructions and a move per direction where a select costs one)
            if (F2_OPT & 2) asm volatile("" : "+v"(v));
            f[K] = (cx == 1 && xl) ? weight(K) * inlet_e : (bb ? own : v);
        } else {
            float v = exch[pl * PLANE - CY[K] * F2_LCOLS - cx + slot];
            if (F2_OPT & 2) asm volatile("" : "+v"(v));
            const bool wall = cx == 1 ? xl : (cx == -1 ? xr : false);
            f[K] = (wall || bb) ? own : v;            // wall bounce-back likewise
        }
    }
};
template <int PRESET>
struct F2<PRESET, -1> {
    template <int WALL> static __device__ __forceinline__ void issue(const char*, const Geo&, unsigned, unsigned, bool, bool, unsigned, unsigned, const F2Wall&) {}
    template <int WALL> static __device__ __forceinline__ void load(const char*, const Geo&, unsigned, unsigned, bool, bool, unsigned, float*, const F2Wall&) {}
    static __device__ __forceinline__ void unstage(const float*, int, bool, float, float*) {}
    static __device__ __forceinline__ void issue_wide(const char*, const Geo&, unsigned, unsigned) {}
    static __device__ __forceinline__ void unstage_wide(const float*, int, float*) {}
    static __device__ __forceinline__ void fix_inlet(bool, float, float*) {}
    static __device__ __forceinline__ void fix_lid(bool, float, float*) {}
    template <int PLANE> static __device__ __forceinline__ void put(float*, int, const float*, const float*, const float*) {}
    template <int PLANE, int WALL> static __device__ __forceinline__ void get(const float*, int, int, bool, bool, float, unsigned, const float*,
                                                                              const float*, const float*, float*, const F2Wall&, float) {}
};

// One block of the list: rows [y0, y0 + tyn) of x tile bx, second steps for the local layers [zs, ze).
// TY: rows per tile.  STAGE 1: the next layer's loads are prefetched into LDS; 0: loaded when needed.
// NBUF 2: two exchange buffers, one barrier per layer; 1: one buffer, two barriers per layer.
// OBST 1: blocks may hold solid cells (an obstacle; not the walls of the domain): directions that would pull
// from a solid cell bounce back (first step: the load address; second step: the own first-step result), solid
// cells are computed like any other and never stored.
// WALL 1 (with OBST 1): see k_fused2 below.
// WIDE 1 (plain blocks of the staged one-buffer variants): the first-step loads in their wide form (F2<>::issue_wide).
// Why: a CU of this kernel is bound by the line requests its L1 can keep in flight (profiles/r03_tuning.md section 6), and
// the halo-column wave's one-sector loads -- 28 lanes x 19 populations, each in a line of its own -- are 532 of a block's
// 1 204 read requests per layer for 2 of its 66 columns.  Loaded as part of the row they sit in they cost the two
// neighbouring lines per row and population instead.  The row waves refill their areas once every wave has taken its
// values out (a third barrier per layer).
template <int PRESET, int TY, int STAGE, int NBUF, int OBST, int WALL, int WIDE>
__device__ __forceinline__ void f2_block(const float* __restrict__ fin, float* __restrict__ fout, const unsigned* __restrict__ nbw,
                                         const Geo& g, const Params& p, int bx, int y0, int tyn, int zs, int ze, bool blk_clean) {
    static_assert(!WALL || OBST, "the wall rules ride on the solid-neighbour bits");
    static_assert(!WIDE || (!OBST && f2_wide_ok(STAGE, NBUF)), "wide loads: plain blocks, staged loads, one exchange buffer");
    constexpr int ROWS = TY + 2;
    constexpr int PLANE = ROWS * F2_LCOLS;
    constexpr int EXCH = 16 * PLANE;
    extern __shared__ float lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int role = f2_role<TY>(wave);
    const int x0 = bx * 64;

    int x, r;
    bool active;
    if (role < ROWS) { r = role; x = x0 + lane; active = true; }
    else { r = lane < ROWS ? lane : lane - ROWS; x = lane < ROWS ? x0 - 1 : x0 + 64; active = lane < 2 * ROWS; }
    const int y = y0 - 1 + r;
    active = active && r <= tyn + 1 && x >= 0 && x < g.W && (!WALL || (y >= 0 && y < g.H));
    const bool second = role >= 1 && role <= tyn && x < g.W;
    // walls: the lane's row is fixed, the layer's z flags change as the block climbs
    const bool w_bot = WALL && y == 0, w_top = WALL && y == g.H - 1;
    const unsigned ybits = WALL ? ((w_bot ? F2_CY_POS : 0u) | (w_top ? F2_CY_NEG : 0u)) : 0u;
    auto wall_of = [&](int L) -> F2Wall {
        const int zg = L - 1 + g.z0;
        return F2Wall{w_bot, w_top, WALL && zg == 0, WALL && zg == g.Dg - 1};
    };
    auto layer_ok = [&](int L) { const int zg = L - 1 + g.z0; return !WALL || (zg >= 0 && zg < g.Dg); };
    const int col = x - x0 + 1;
    const int slot = r * F2_LCOLS + col;
    const bool xl = x == 0, xr = x == g.W - 1;
    const int slotm = slot - (xr ? 1 : 0);                       // outlet lane: its cx = -1 directions copy from its own column
    const float inlet_e = inlet_bracket(p.U);
    float* stage = lds + NBUF * EXCH + wave * (Q * (WIDE ? F2_WIDE : 64));
    // WIDE: where the lane finds its cell's staged values: the area of the wave that holds its row, at its column
    const float* wide_src = nullptr;
    if (WIDE) {
        const int src_wave = role < ROWS ? wave : f2_wave_of<TY>(r);
        wide_src = lds + NBUF * EXCH + src_wave * (Q * F2_WIDE) + (x - (x0 - 2));
    }
    const bool row_wave = role < ROWS;
    // the staging area's LDS byte address for the LDS-DMA loads (M0).  Inside the layer loop it passes through an empty
    // asm, like the plane stride of the stores: derived per iteration by scalar adds, the 19 M0 values and 19 store
    // bases do not occupy 57 scalar registers across the loop (there are none to spare: they spilled into vector lanes)
    unsigned stage_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)stage;

    // input addressing: byte offsets from layer zs - 3 (two layers of slack keep every offset, the wall and
    // outlet adjustments included, non-negative)
    const char* base_in = reinterpret_cast<const char*>(fin + (long long)(zs - 3) * g.layer);
    char* base_out = reinterpret_cast<char*>(fout + (long long)(zs - 3) * g.layer);
    const unsigned layer_b = (unsigned)(g.layer * 4), plane_b = (unsigned)(g.plane * 4);
    const unsigned cell_b = (unsigned)(((long long)(active ? y : y0) * g.row + (active ? x : x0)) * 4);
    auto off_of = [&](int L) { return (unsigned)(L - (zs - 3)) * layer_b + cell_b; };
    // WIDE: lane i < 17 of a row wave loads the 16 bytes at float x0 - 2 + 4 i of its row
    // (whatever the lane's own cell: a tile at the ragged end of a row still loads its row's 68 floats -- the padding and
    // the next row's start are in bounds and never used)
    const bool row_on = row_wave && role <= tyn + 1;
    const unsigned wide_b = (unsigned)(((long long)(row_on ? y0 - 1 + role : y0) * g.row + x0) * 4 + lane * 16 - 8);
    auto offw_of = [&](int L) { return (unsigned)(L - (zs - 3)) * layer_b + wide_b; };
    const bool wide_loader = WIDE && row_on && lane < 17;

    // solid-neighbour bits of the lane's cell in layer L (0 without OBST: everything below folds away)
    const unsigned nb_cell = (unsigned)((active ? y : y0) * g.Wp + (active ? x : x0));
    const unsigned* nb_base = nbw + (long long)(zs - 3) * g.H * g.Wp;
    auto nb_of = [&](int L) -> unsigned {
        if (!OBST || !active) return 0u;
        unsigned wb = 0u;
        if (WALL) {
            const int zg = L - 1 + g.z0;
            wb = ybits | (zg == 0 ? F2_CZ_POS : 0u) | (zg == g.Dg - 1 ? F2_CZ_NEG : 0u);
        }
        if (blk_clean) return wb;
        return nb_base[(unsigned)(L - (zs - 3)) * (unsigned)(g.H * g.Wp) + nb_cell] | wb;
    };

    float below[Q], cur[Q], above[Q], pf[Q];   // first-step results of layers L-2, L-1, L; the loads of layer L
#pragma unroll
    for (int k = 0; k < Q; k++) { below[k] = 0.f; cur[k] = 0.f; above[k] = 0.f; pf[k] = 0.f; }
    // bits of layers L-1, L, L+1: loaded two layers before the first use (the addresses of the first-step loads)
    unsigned nb_cur = 0u, nb_above = nb_of(zs - 1), nb_next = nb_of(zs);
    if (STAGE) {
        const unsigned off = off_of(zs - 1);
        if (WIDE) {
            if (wide_loader) F2<PRESET, Q - 1>::issue_wide(base_in, g, offw_of(zs - 1), stage_lds);
        } else if (active && layer_ok(zs - 1) && (!OBST || __ballot(active && !(nb_above & 1u)) != 0ull))
            F2<PRESET, Q - 1>::template issue<WALL>(base_in, g, off, off - (xr ? 4u : 0u), xl, xr, nb_above, stage_lds, wall_of(zs - 1));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (WIDE) __syncthreads();
    }
    int buf = 0;
    unsigned plane_s = plane_b;
    for (int L = zs - 1; L <= ze; L++) {
        if (F2_OPT & 8) asm volatile("" : "+s"(stage_lds), "+s"(plane_s));
        // ---- first step of layer L ---------------------------------------------------------------------
        // (OBST: a wave whose cells of this layer are all solid neither loads nor collides: wave-uniform)
        // (WALL: a layer outside the domain -- below its first, above its last -- has no first step)
        if (WIDE) {
            // every wave takes its cells' values out of the row waves' areas; once all have (one more barrier, right
            // behind the previous layer's last one), the row waves refill them: a whole layer ahead of their use
            if (active) F2<PRESET, Q - 1>::unstage_wide(wide_src, xr ? 1 : 0, pf);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (wide_loader && L < ze) F2<PRESET, Q - 1>::issue_wide(base_in, g, offw_of(L + 1), stage_lds);
            if (active) {
                F2<PRESET, Q - 1>::fix_inlet(xl, inlet_e, pf);
                F2_COLLIDE(pf, above);
            }
        } else if (active && layer_ok(L) && (!OBST || __ballot(active && !(nb_above & 1u)) != 0ull)) {
            if (STAGE) {
                F2<PRESET, Q - 1>::unstage(stage, lane, xl, inlet_e, pf);
            } else {
                const unsigned off = off_of(L);
                F2<PRESET, Q - 1>::template load<WALL>(base_in, g, off, off - (xr ? 4u : 0u), xl, xr, nb_above, pf, wall_of(L));
            }
            F2<PRESET, Q - 1>::fix_inlet(xl, inlet_e, pf);
            if (WALL) F2<PRESET, Q - 1>::fix_lid(w_top, p.U, pf);
            F2_COLLIDE(pf, above);
        }
        if (STAGE && !WIDE) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging area has been read: refill it
            if (active && L < ze && layer_ok(L + 1) && (!OBST || __ballot(active && !(nb_next & 1u)) != 0ull)) {
                const unsigned off = off_of(L + 1);
                // (most waves have no solid cell or wall in reach: they take the plain form of the loads, wave-uniformly)
                if (!OBST || __ballot(nb_next != 0u) == 0ull) F2<PRESET, Q - 1>::template issue<0>(base_in, g, off, off - (xr ? 4u : 0u), xl, xr, 0u, stage_lds, F2Wall{});
                else F2<PRESET, Q - 1>::template issue<WALL>(base_in, g, off, off - (xr ? 4u : 0u), xl, xr, nb_next, stage_lds, wall_of(L + 1));
            }
        }
        const unsigned nb_next2 = L + 2 <= ze ? nb_of(L + 2) : 0u;
        // ---- second step of layer L - 1 ------------------------------------------------------------------
        if (L >= zs + 1) {
            float* exch = lds + buf * EXCH;
            if (active) F2<PRESET, Q - 1>::template put<PLANE>(exch, slot, below, cur, above);
            __syncthreads();
            if (second && (!OBST || __ballot(second && !(nb_cur & 1u)) != 0ull)) {
                float f[Q], o[Q];
                if (!OBST || __ballot(nb_cur != 0u) == 0ull)
                    F2<PRESET, Q - 1>::template get<PLANE, 0>(exch, slot, slotm, xl, xr, inlet_e, 0u, below, cur, above, f, F2Wall{}, p.U);
                else
                    F2<PRESET, Q - 1>::template get<PLANE, WALL>(exch, slot, slotm, xl, xr, inlet_e, nb_cur, below, cur, above, f, wall_of(L - 1), p.U);
                F2_COLLIDE(f, o);
                // the staged loads have had the whole second step to land; waiting for them HERE keeps the
                // latency of the stores below out of the next iteration (stores count in vmcnt too)
                if (STAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // a scalar base per population (layer and plane offsets: scalar adds) + the lane's constant 32-bit
                // cell offset: no vector instruction spent on store addresses
                char* out_layer = base_out + (size_t)((unsigned)(L - 1 - (zs - 3)) * layer_b);
                const unsigned offo = off_of(L - 1);
                if (!OBST || !(nb_cur & 1u)) {
#pragma unroll
                    for (int k = 0; k < Q; k++) {
                        if (F2_OPT & 4) __builtin_nontemporal_store(o[k], reinterpret_cast<float*>(out_layer + (size_t)((unsigned)k * plane_s) + cell_b));
                        else __builtin_nontemporal_store(o[k], reinterpret_cast<float*>(base_out + (offo + (unsigned)k * plane_b)));
                    }
                }
            } else {
                if (STAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (NBUF == 1) __syncthreads(); else buf ^= 1;
        } else {
            if (STAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (WIDE) __syncthreads();   // (the two lead-in layers of a block have no second step and so no barrier of their own)
        }
#pragma unroll
        for (int k = 0; k < Q; k++) { below[k] = cur[k]; cur[k] = above[k]; }
        nb_cur = nb_above; nb_above = nb_next; nb_next = nb_next2;
    }
}

// WALL 1 (with OBST 1): blocks may lie on the domain's y and z walls; the rows / layers outside the domain take no part,
// and the wall rules ride on the solid-neighbour bits (F2Wall).  The wind tunnel's outlet tile column keeps off the
// walls (host side): its outlet branch indexes past the row / layer there (quirk Q2) into cells no block holds.
template <int PRESET, int TY, int STAGE, int NBUF, int OBST, int WALL>
__global__ __launch_bounds__((TY + 3) * 64) void k_fused2(const float* __restrict__ fin, float* __restrict__ fout,
                                                          const unsigned long long* __restrict__ blocks,
                                                          const unsigned* __restrict__ nbw, Geo g, Params p,
                                                          FusedGrid fg, unsigned long long* __restrict__ stamps, F2Rim rim) {
    if (blockIdx.x >= (unsigned)fg.nblocks) {   // behind the block list: a step of the rim, one tile per wave
        const unsigned idx = (blockIdx.x - (unsigned)fg.nblocks) * (unsigned)(TY + 3) + (threadIdx.x >> 6);
        if (idx < rim.n) f2_rim_tile<PRESET>(rim, g, p, rim.list[idx], threadIdx.x & 63);
        return;
    }
    // One workgroup per block of the host's list (fused_prepare() in lbm_hip.hip: a 64 x TY tile and a run of
    // layers whose haloed neighbourhood is plain fluid, ordered layer range / y tile / x tile; the rim path owns
    // every other cell).  The hardware deals workgroups to the 8 XCDs round-robin in launch
    // order, and each XCD has its own L2.  Neighbouring tiles read each other's edge lines (the halo rows and
    // columns, the second 128-byte line of an x-shifted load), so every XCD takes ONE contiguous range of the
    // list: the lines two tiles share are then fetched from HBM once instead of once per L2, and every XCD
    // still gets the same number of blocks whatever the obstacles leave eligible.
    int bx, y0, tyn, zs, ze;                                     // rows [y0, y0 + tyn); second step: local layers [zs, ze)
    bool blk_clean, blk_plain;
    {
        // blocks[0..8]: where each XCD's range of the list starts (ranges of equal WORK, not equal count: the
        // runs an obstacle leaves differ in length); workgroup n of XCD x takes entry n of range x, a workgroup
        // beyond its range has nothing to do
        const unsigned xcd = blockIdx.x & 7u, nth = blockIdx.x >> 3;
        const unsigned first = (unsigned)blocks[xcd], end = (unsigned)blocks[xcd + 1];
        if (nth >= end - first) return;
        const unsigned long long e = blocks[9 + first + nth];
        // the entry is the same for the whole workgroup: keep what is derived from it in scalar registers
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)e), hi = __builtin_amdgcn_readfirstlane((unsigned)(e >> 32));
        // x tile : 12 | first row : 16 | rows : 4 | first second-step layer : 16 | layers : 14 | plain : 1 | clean : 1
        bx = (int)(lo & 0xfffu);
        y0 = (int)((lo >> 12) & 0xffffu);
        tyn = (int)(lo >> 28);
        zs = (int)(hi & 0xffffu);
        // (the host never lets a block run past zl_end; written as a min() the compiler cannot derive a minimum trip
        // count of the layer loop from the entry's layout, which made it peel the loop and spill 18 registers)
        ze = min(zs + (int)((hi >> 16) & 0x3fffu), fg.zl_end);
        // bit 63: no solid cell within reach of the block (OBST form: it then reads no neighbour words at all)
        blk_clean = (hi >> 31) != 0u;
        // bit 62: neither a solid cell nor a wall of the domain within reach: the block runs the plain form of the loop
        // inside the obstacle / wall forms of the kernel too (those cost 1.7 % / 2.5 % on a block that needs neither)
        blk_plain = ((hi >> 30) & 1u) != 0u;
    }
    if (stamps && threadIdx.x == 0) stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    // plain blocks of the staged one-buffer variants load wide (the cavity's wall columns bounce back per lane: narrow)
    constexpr bool CAN_WIDE = F2_WIDE_LOADS && f2_wide_ok(STAGE, NBUF);
    const bool plain = !OBST || blk_plain;
    const bool wide = CAN_WIDE && plain && fg.wide && (PRESET == LBM_PRESET_WIND_TUNNEL_REF || (bx > 0 && bx < g.Wt - 1));
    if (CAN_WIDE && wide) f2_block<PRESET, TY, STAGE, NBUF, 0, 0, CAN_WIDE ? 1 : 0>(fin, fout, nbw, g, p, bx, y0, tyn, zs, ze, true);
    else if (plain) f2_block<PRESET, TY, STAGE, NBUF, 0, 0, 0>(fin, fout, nbw, g, p, bx, y0, tyn, zs, ze, true);
    else f2_block<PRESET, TY, STAGE, NBUF, OBST, WALL, 0>(fin, fout, nbw, g, p, bx, y0, tyn, zs, ze, blk_clean);
    if (stamps && threadIdx.x == 0) stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}
