// k_step.h -- geometry, boundary classification and the fused stream-collide kernel (A/B pull scheme).
// Part of the single translation unit lbm_hip.hip (included inside its anonymous namespace, after
// <hip/hip_runtime.h>, include/lbm_abi.h and d3q19.h); not a stand-alone header.
// ------------------------------------------------------------------------------------------
// geometry shared by host and device
// ------------------------------------------------------------------------------------------
struct Geo {
    int W, H, D;        // owned extent of this context
    int Dg, z0;         // global depth, global z of owned layer 0
    int slab;           // 1: one z-slab of a taller domain (ghost layers are fed by the halo exchange)
    int fz0, fnz;       // field window: rho/u are kept for owned layers [fz0, fz0+fnz) only
    int Wp;             // floats per row of one population (multiple of 32: rows start on 128-byte lines)
    int Wm;             // 64-bit mask words per row
    int Wt;             // 64-cell tiles per row (= Wm)
    long long row;      // floats between consecutive y rows of one population
    long long layer;    // floats between consecutive z layers of one population
    long long plane;    // floats between consecutive populations of one cell
    long long mlayer;   // floats per population layer inside a halo message (H*Wp)
};
// Two layouts share every kernel through these strides:
//   LBM_LAYOUT_PLANES  [k][z][y][x] : row = Wp,    layer = rows*Wp,  plane = (D+2)*layer + skew
//   LBM_LAYOUT_ROWS    [z][y][k][x] : plane = Wp,  row = 19*Wp,      layer = rows*row
// In the row-interleaved form the 19 runs of one (y,z) row are one contiguous 19*Wp*4-byte chunk:
// a step writes whole chunks and reads 1-3 KB pieces of nine neighbouring chunks, which streams
// 6-7 % faster from HBM than 19+19 far-apart planes (profiles/r01_tuning.md, tools/probes/membench5/6).

struct Params {
    float U, omega_shear, omega_bulk;
};

// device address of cell (x, y, zl) inside a plane; zl = local layer index incl. ghost (owned: 1..D)
__device__ __forceinline__ long long cell_off(const Geo& g, int x, int y, int zl) {
    return (long long)zl * g.layer + (long long)y * g.row + x;
}

// flat thread index of the 1-D kernels.  A dispatch holds at most 2^32 - 1 work-items per grid
// dimension, and a buffer sized to HBM has more floats than that, so those kernels are launched on
// the 2-D grid of grid1() and flatten it here.
__device__ __forceinline__ long long gidx() {
    return ((long long)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
}

__device__ __forceinline__ bool solid_at(const Geo& g, const unsigned long long* __restrict__ mask,
                                         int x, int y, int zl) {
    const unsigned long long w = mask[((long long)zl * g.H + y) * g.Wm + (x >> 6)];
    return (w >> (x & 63)) & 1ull;
}

// Populations whose A/B fast-path load is non-temporal when TUNE bit 0 is set (bit k = population k).
// Default: the ten x-shifted populations keep the default policy -- their misaligned loads touch a
// second 128-byte line that the neighbouring wave needs too; nt would drop it after one use
// (+2 % at 256^3, tools/probes/tune_bench.py).
#ifndef LBM_NT_SET
#define LBM_NT_SET 0x78079u   // cx == 0: k = 0, 3, 4, 5, 6, 15, 16, 17, 18
#endif
// population load / store with an explicit cache policy (TUNE bit 0: nt loads, bit 1: sc1 stores)
template <int TUNE>
__device__ __forceinline__ float ldf(const float* p) {
    return (TUNE & 1) ? __builtin_nontemporal_load(p) : *p;
}
template <int TUNE, int K>
__device__ __forceinline__ float ldk(const float* p) {
    return ((TUNE & 1) && ((LBM_NT_SET >> K) & 1u)) ? __builtin_nontemporal_load(p) : *p;
}
template <int TUNE>
__device__ __forceinline__ void stf(float* p, float v) {
    if (TUNE & 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dword ... sc1
    else __builtin_nontemporal_store(v, p);
}

// Which branch of the reference chain (lbm_solver.cu:283-318) feeds direction K of the fluid cell
// (x, y, zl).  Single source of truth for the step kernel, the moment kernel and
// lbm_read_classification.
// `nb_solid()` answers "is the cell this direction pulls from solid?"; it is only asked when that
// cell exists.
template <int PRESET, int K, class NbSolid>
__device__ __forceinline__ unsigned classify_with(const Geo& g, int x, int y, int zl, NbSolid nb_solid) {
    constexpr int cx = CX[K], cy = CY[K], cz = CZ[K];
    const int nx = x - cx, ny = y - cy;
    const int nzg = (zl - 1) + g.z0 - cz;
    if (PRESET == LBM_PRESET_CAVITY) {
        if (cy == -1 && ny >= g.H) return SRC_LID_BB;
        if ((cx == 1 && nx < 0) || (cx == -1 && nx >= g.W) || (cy == 1 && ny < 0) ||
            (cz != 0 && (nzg < 0 || nzg >= g.Dg)))
            return SRC_WALL_BB;
    } else {
        if (cx == 1 && nx < 0) return SRC_INLET;
        if (cx == -1 && nx >= g.W) return SRC_OUTLET;
        if (cy == 1 && ny < 0) return SRC_FLOOR_BB;
        if (cy == -1 && ny >= g.H) return SRC_CEIL_RFY;
        if (cz != 0 && (nzg < 0 || nzg >= g.Dg)) return SRC_SIDE_RFZ;
    }
    if (K == 0) return SRC_PULL;
    return nb_solid() ? SRC_SOLID_BB : SRC_PULL;
}

// per-cell form: the neighbour's bit is read from the packed mask in memory
template <int PRESET, int K>
__device__ __forceinline__ unsigned classify_dir(const Geo& g, const unsigned long long* __restrict__ mask,
                                                 int x, int y, int zl) {
    return classify_with<PRESET, K>(g, x, y, zl, [&] { return solid_at(g, mask, x - CX[K], y - CY[K], zl - CZ[K]); });
}

// Per-tile form used by the step kernels: the solid bits of the 3 x 3 rows around one 64-cell
// tile are fetched ONCE into wave-uniform registers (scalar loads), so the 19 directions are
// classified without a memory round trip each and the 19 population loads issue back to back.
struct TileMask {
    unsigned long long w[3][3];  // [dz+1][dy+1]: cells x0 .. x0+63 of row (y+dy, zl+dz); 0 where that row does not exist
    unsigned left, right;        // bit 3*(dz+1)+(dy+1): cell x0-1 / x0+64 of the same row
};
__device__ __forceinline__ TileMask load_tile_mask(const Geo& g, const unsigned long long* __restrict__ mask,
                                                   int xs, int y, int zl) {
    TileMask t;
    t.left = 0; t.right = 0;
#pragma unroll
    for (int dz = -1; dz <= 1; dz++)
#pragma unroll
        for (int dy = -1; dy <= 1; dy++) {
            const int yy = y + dy;
            const bool ok = yy >= 0 && yy < g.H;  // layers zl-1 .. zl+1 always exist (ghost layers)
            const unsigned long long* r = mask + ((long long)(zl + dz) * g.H + (ok ? yy : y)) * g.Wm;
            const unsigned long long w = r[xs];
            const unsigned long long wl = r[xs > 0 ? xs - 1 : xs];
            const unsigned long long wr = r[xs + 1 < g.Wm ? xs + 1 : xs];
            t.w[dz + 1][dy + 1] = ok ? w : 0ull;
            const int bit = 3 * (dz + 1) + (dy + 1);
            if (ok && xs > 0) t.left |= (unsigned)(wl >> 63) << bit;
            if (ok && xs + 1 < g.Wm) t.right |= (unsigned)(wr & 1ull) << bit;
        }
    return t;
}
template <int K>
__device__ __forceinline__ bool tile_nb_solid(const TileMask& t, int lane) {
    constexpr int iz = 1 - CZ[K], iy = 1 - CY[K], bit = 3 * iz + iy;
    unsigned long long w = t.w[iz][iy];
    if (CX[K] == 1) w = (w << 1) | (unsigned long long)((t.left >> bit) & 1u);           // neighbour x-1
    if (CX[K] == -1) w = (w >> 1) | ((unsigned long long)((t.right >> bit) & 1u) << 63);  // neighbour x+1
    return (w >> lane) & 1ull;
}

// Outlet branch (lbm_solver.cu:292-296): the reference indexes f_in[k*n + nz*W*H + ny*W + (W-1)]
// without range-checking ny/nz, so a row off the y end carries into the next z layer and a layer
// off the z end carries into the neighbouring population plane (quirk Q2).  Our planes are padded,
// so the carry is applied in logical coordinates.
template <int K>
__device__ __forceinline__ long long outlet_off(const Geo& g, int y, int zl) {
    int pl = K;
    int ny = y - CY[K];
    int nzg = (zl - 1) + g.z0 - CZ[K];
    if (ny < 0) { ny += g.H; nzg -= 1; }
    else if (ny >= g.H) { ny -= g.H; nzg += 1; }
    // An undecomposed domain holds the wrapped layer itself; a slab finds it in its ghost layer,
    // where the ring halo exchange put the neighbour's values (see k_halo_unpack).
    if (nzg < 0) { pl -= 1; if (!g.slab) nzg += g.Dg; }
    else if (nzg >= g.Dg) { pl += 1; if (!g.slab) nzg -= g.Dg; }
    const int nzl = nzg - g.z0 + 1;
    return (long long)pl * g.plane + cell_off(g, g.W - 1, ny, nzl);
}
template <int K>
__device__ __forceinline__ float outlet_read(const Geo& g, const float* __restrict__ fin, int y, int zl) {
    return fin[outlet_off<K>(g, y, zl)];
}

// Outlet column side array `ob` (in-place schemes).  The outlet branch copies from another FLUID cell
// of the x = W-1 column (and, through quirk Q2, from the far end of the domain), which an in-place
// update would race with or have overwritten already: the seven planes it can touch are mirrored
// for that column in a small double-buffered array that x = W-1 cells write each step and read the
// step after.
constexpr int OB_N = 7;
constexpr int OB_PLANE[OB_N] = {2, 8, 9, 10, 12, 13, 14};
__host__ __device__ constexpr int ob_index(int pl) { return pl == 2 ? 0 : (pl <= 10 ? pl - 7 : pl - 8); }
// side-array slot of plane-slot i, local layer z in [-1, D] (ghost rows for the z-slab halo), row y
__device__ __forceinline__ long long ob_at(const Geo& g, int i, int z, int y) {
    return ((long long)i * (g.D + 2) + (z + 1)) * g.H + y;
}

// the in-place form of outlet_read(): same carry rules, value taken from the side array
template <int K>
__device__ __forceinline__ float aa_outlet_read(const Geo& g, const float* __restrict__ ob_old, int y, int zl) {
    int pl = K;
    int ny = y - CY[K];
    int nz = (zl - 1) - CZ[K];
    if (ny < 0) { ny += g.H; nz -= 1; }
    else if (ny >= g.H) { ny -= g.H; nz += 1; }
    // as outlet_off(): an undecomposed domain wraps to its own far layer, a slab finds the
    // neighbour's (or, over the ring link, the far rank's) column in its ghost row
    const int nzg = nz + g.z0;
    if (nzg < 0) { pl -= 1; if (!g.slab) nz += g.D; }
    else if (nzg >= g.Dg) { pl += 1; if (!g.slab) nz -= g.D; }
    return ob_old[ob_at(g, ob_index(pl), nz, ny)];
}

// full boundary chain for one direction, per lane: the branch is turned into an address by
// selects, so every lane issues exactly one load and no load waits for another
template <int PRESET, int K, int OB>
__device__ __forceinline__ float gather_general(const Geo& g, const float* __restrict__ fin, const TileMask& t,
                                                int lane, int x, int y, int zl, long long own, float inlet_e, float U,
                                                const float* __restrict__ ob_old) {
    const unsigned code = classify_with<PRESET, K>(g, x, y, zl, [&] { return tile_nb_solid<K>(t, lane); });
    // own-cell sources: ceiling / side mirror, bounce-back (floor, solid, wall, lid); an inlet lane
    // loads its own slot and discards it
    const int pl = code == SRC_CEIL_RFY ? RFY[K] : (code == SRC_SIDE_RFZ ? RFZ[K] : (code == SRC_INLET ? K : INV[K]));
    long long a = (long long)pl * g.plane + own;
    if (code == SRC_PULL) a = (long long)K * g.plane + own - ((long long)CZ[K] * g.layer + (long long)CY[K] * g.row + CX[K]);
    // sliding in-place scheme, undecomposed: the wrapped outlet index (Q2) would read layers the sweep
    // has overwritten, so the column's values of the previous step come from the side array.  A slab
    // finds the wrapped layer in its ghost layer like A/B does.
    const bool side_array = OB && !g.slab;
    if (PRESET == LBM_PRESET_WIND_TUNNEL_REF && CX[K] == -1 && !side_array) {
        const long long ao = outlet_off<K>(g, y, zl);  // wave-uniform
        if (code == SRC_OUTLET) a = ao;
    }
    float v = fin[a];
    if (PRESET == LBM_PRESET_WIND_TUNNEL_REF && CX[K] == -1 && side_array) {
        // (an outlet lane loaded its own slot above and discards it)
        if (code == SRC_OUTLET) v = aa_outlet_read<K>(g, ob_old, y, zl);
    }
    if (PRESET == LBM_PRESET_WIND_TUNNEL_REF && CX[K] == 1 && code == SRC_INLET) v = weight(K) * inlet_e;
    if (PRESET == LBM_PRESET_CAVITY && code == SRC_LID_BB) v = v + lid_term(K, U);
    return v;
}

// interior tile: rows y-1..y+1 and layers z-1..z+1 exist, no solid cell within reach; only the
// x ends of the row need care.
template <int PRESET, int K, int TUNE>
__device__ __forceinline__ float gather_fast(const Geo& g, const float* __restrict__ fin,
                                             int x, int y, int zl, long long own, float inlet_e) {
    constexpr int cx = CX[K];
    const float* row = fin + (long long)K * g.plane + (long long)(zl - CZ[K]) * g.layer +
                       (long long)(y - CY[K]) * g.row;
    if (cx == 0) return ldk<TUNE, K>(row + x);
    if (PRESET == LBM_PRESET_WIND_TUNNEL_REF) {
        if (cx == 1) {
            const float v = ldk<TUNE, K>(row + x - 1);  // x = 0 reads the pad/previous row: in bounds, discarded
            return x == 0 ? weight(K) * inlet_e : v;
        }
        const int xs = x + 1 < g.W ? x + 1 : g.W - 1;  // outlet: copy from x = W-1 (ny, nz in range here)
        return ldk<TUNE, K>(row + xs);
    } else {
        const bool wall = (cx == 1) ? (x == 0) : (x == g.W - 1);
        const float* p = wall ? fin + (long long)INV[K] * g.plane + own : row + x - cx;  // one load either way
        return ldk<TUNE, K>(p);
    }
}

template <int PRESET, int K>
struct GatherAll {
    template <int TUNE>
    static __device__ __forceinline__ void fast(const Geo& g, const float* __restrict__ fin, int x, int y,
                                                int zl, long long own, float inlet_e, float* f) {
        GatherAll<PRESET, K - 1>::template fast<TUNE>(g, fin, x, y, zl, own, inlet_e, f);
        f[K] = gather_fast<PRESET, K, TUNE>(g, fin, x, y, zl, own, inlet_e);
    }
    template <int OB>
    static __device__ __forceinline__ void general(const Geo& g, const float* __restrict__ fin, const TileMask& t,
                                                   int lane, int x, int y, int zl, long long own, float inlet_e,
                                                   float U, const float* __restrict__ ob_old, float* f) {
        GatherAll<PRESET, K - 1>::template general<OB>(g, fin, t, lane, x, y, zl, own, inlet_e, U, ob_old, f);
        f[K] = gather_general<PRESET, K, OB>(g, fin, t, lane, x, y, zl, own, inlet_e, U, ob_old);
    }
    static __device__ __forceinline__ void classify(const Geo& g, const unsigned long long* __restrict__ mask,
                                                    int x, int y, int zl, unsigned char* out) {
        GatherAll<PRESET, K - 1>::classify(g, mask, x, y, zl, out);
        out[K] = (unsigned char)classify_dir<PRESET, K>(g, mask, x, y, zl);
    }
};
template <int PRESET>
struct GatherAll<PRESET, -1> {
    template <int TUNE>
    static __device__ __forceinline__ void fast(const Geo&, const float*, int, int, int, long long, float, float*) {}
    template <int OB>
    static __device__ __forceinline__ void general(const Geo&, const float*, const TileMask&, int, int, int, int,
                                                   long long, float, float, const float*, float*) {}
    static __device__ __forceinline__ void classify(const Geo&, const unsigned long long*, int, int, int,
                                                    unsigned char*) {}
};

// Tile -> (xs, y, z) mapping shared by the kernels that walk tiles: a 256-thread block is four
// waves; `segsx` of them sit side by side in x, the rest stack in y.
// The hardware deals workgroups out to the 8 XCDs round-robin in launch order (x fastest), so the
// workgroups that sit side by side in one row land on different XCDs and the cache line their
// x-shifted loads share at the seam is fetched into two L2s.  With ZORDER_XCD the launch order is
// re-read so that each XCD gets whole rows: of every 8 consecutive rows of workgroups, row j goes to XCD j.
// ZORDER_PRIO: the launch runs beside k_fused2 (the boundary layers and the rim of a pair of steps) and something waits
// for it: its waves raise their issue priority over the long kernel's (s_setprio)
// ZORDER_WAVE: workgroups of ONE wave (launched with 64 threads, grid = segments x rows x layers).  For the boundary
// layers' second step of a pair, which runs beside k_fused2: that kernel's 15 waves per CU leave the registers of three
// more waves of this one on a CU (118 of 512 per SIMD lane times 4 + 4 + 4 + 3 waves), but not of a workgroup of four,
// which therefore waited until the fused launch was over (profiles/r03_tuning.md section 12).
constexpr int ZORDER_REVERSE = 1, ZORDER_XCD = 2, ZORDER_PRIO = 4, ZORDER_WAVE = 8;
struct TileId {
    int xs, y, lane, bz;
};
__device__ __forceinline__ TileId tile_of_wave(int segsx_log2, int zorder) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TileId t;
    t.lane = threadIdx.x & 63;
    const int segsx = 1 << segsx_log2;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (zorder & ZORDER_WAVE) {
        t.xs = bx; t.y = by; t.bz = bz;
        return t;
    }
    if ((zorder & ZORDER_XCD) && gridDim.x > 1) {
        const unsigned gx = gridDim.x, gy = gridDim.y;
        const unsigned rows = gy * gridDim.z;                     // rows of workgroups in this launch
        const unsigned L = bx + gx * (by + gy * bz);              // launch order
        const unsigned group = L / (8 * gx), within = L - group * (8 * gx);
        if (group * 8 + 8 <= rows) {                              // the ragged tail keeps the launch order
            const unsigned R = group * 8 + (within & 7);
            bx = within >> 3;
            bz = R / gy;
            by = R - bz * gy;
        }
    }
    t.xs = bx * segsx + (wave & (segsx - 1));
    t.y = by * (4 >> segsx_log2) + (wave >> segsx_log2);
    t.bz = bz;
    return t;
}

// ------------------------------------------------------------------------------------------
// the hot kernel: fused pull-stream + boundary chain + collision (lbm_solver.cu:242-402)
// MODE 0: step (writes f_out; with want_moments also rho/u of the field window, fused: 20 B/cell
//         instead of the 96 B/cell of a MODE 1 pass).  MODE 1: moments only (writes rho/u in the
//         reference layout).
// MODE 2: step of the one or two slab-boundary layers (blockIdx.z = 0 -> zl_begin, 1 -> zl_second)
//         that also writes the outgoing halo messages, so no separate pack pass is needed.
// MODE 3: step of the sliding in-place scheme (LBM_SCHEME_SLIDE): fin and fout are the SAME buffer
//         at different layer offsets, launched chunk by chunk (see slide_step() on the host); the
//         outlet branch reads the side array, x = W-1 cells write it, and rho/u are written here
//         when asked for (there is no intact previous buffer to recompute them from).  As a z-slab it
//         writes the halo messages of the boundary layers like MODE 2; with fout == nullptr it ONLY
//         writes messages (the boundary layer that the directed sweep reaches last must send early).
// MODE 4: MODE 0 over an explicit list of tiles instead of a box of layers (one entry per wave): the rim of
//         the two-steps-per-pass scheme (k_fused2.h), i.e. the tiles the fused kernel does not cover.
// ------------------------------------------------------------------------------------------
// message layout: see k_halo_pack
__device__ __forceinline__ void emit_halo(const Geo& g, float* __restrict__ msg, int side, int x, int y,
                                          const float* v) {
    const long long at = (long long)y * g.Wp + x;
#pragma unroll
    for (int j = 0; j < 5; j++) msg[(long long)j * g.mlayer + at] = v[side ? UP5[j] : DOWN5[j]];
    if (x != g.W - 1) return;
    float* tail = msg + 5 * g.mlayer;
    if (side) {
        if (y == g.H - 1) { tail[0] = v[10]; tail[1] = v[9]; }
        tail[2 + y] = v[13];
    } else if (y == 0) {
        tail[0] = v[8];
        tail[1] = v[9];
    }
}

template <int PRESET, int MODE, int TUNE>
__global__ __launch_bounds__(256) void k_stream_collide(
    const float* __restrict__ fin, float* __restrict__ fout,
    const unsigned long long* __restrict__ mask, const unsigned char* __restrict__ flags,
    float* __restrict__ rho_out, float4* __restrict__ u_out,
    Geo g, Params p, int zl_begin, int segsx_log2, int zrev,
    int zl_second, float* __restrict__ msg_bottom, float* __restrict__ msg_top,
    const float* __restrict__ ob_old, float* __restrict__ ob_new, int want_moments,
    const unsigned long long* __restrict__ tile_list) {
    TileId t;
    int zl;
    if ((MODE == 2 || MODE == 4) && (zrev & ZORDER_PRIO)) __builtin_amdgcn_s_setprio(3);
    if (MODE == 4) {
        // one list entry per wave: xs | y << 16 | zl << 40 (host side: fused_prepare() in lbm_hip.hip); ~0 pads the list
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const unsigned long long e = tile_list[(long long)blockIdx.x * 4 + wave];
        if (e == ~0ull) return;
        t.xs = (int)(e & 0xffffu); t.y = (int)((e >> 16) & 0xffffffu); t.lane = threadIdx.x & 63; t.bz = 0;
        zl = (int)(e >> 40);
        want_moments = 0;
    } else {
        t = tile_of_wave(segsx_log2, zrev);
        // local layer incl. ghost offset (owned: 1..D); zrev walks the layers top-down so that a step
        // starts on the layers the previous step wrote last (still in the Infinity Cache)
        zl = MODE == 2 ? (t.bz == 0 ? zl_begin : zl_second)
                       : zl_begin + ((zrev & ZORDER_REVERSE) ? (int)(gridDim.z - 1 - t.bz) : t.bz);
    }
    const int y = t.y;
    if (t.xs >= g.Wt || y >= g.H) return;
    const int x = t.xs * 64 + t.lane;
    const unsigned char flag = flags[((long long)(zl - 1) * g.H + y) * g.Wt + t.xs];
    if (x >= g.W) return;
    // MODE 3, want_moments bit 1: every solid cell still holds the rest populations init() wrote
    // (w_k), so they need not travel with the lattice: solid lanes neither load nor store (but for the
    // outlet column, which a slab's outlet branch reads from the lattice) and all-solid tiles exit
    const bool skip_solid = MODE == 3 && (want_moments & 2);
    // want_moments bit 2 (MODE 0): nobody reads rho / u before the next step rewrites them (lbm_step_frames without the
    // curl pass): store them past the caches like the populations (256^3: 487 -> 480 us per frame; with a reader right
    // behind -- the curl pass -- the plain store is the faster one, 557 against 560 us)
    const bool nt_fields = MODE == 0 && (want_moments & 4);
    if (MODE == 0 || MODE == 3) want_moments = (want_moments & 1) && (unsigned)(zl - 1 - g.fz0) < (unsigned)g.fnz;
    if ((MODE == 0 || MODE == 4) && flag == 2 && !want_moments) return;  // every cell of the tile is solid: nothing moves
    if (MODE == 3 && skip_solid && flag == 2 && !want_moments && t.xs != g.Wt - 1 &&
        !((zl == 1 && msg_bottom) || (zl == g.D && msg_top)))
        return;

    const long long own = cell_off(g, x, y, zl);
    const float inlet_e = inlet_bracket(p.U);
    float f[Q];

    if (flag == 0) {
        GatherAll<PRESET, Q - 1>::template fast<TUNE>(g, fin, x, y, zl, own, inlet_e, f);
    } else {
        const TileMask tm = load_tile_mask(g, mask, t.xs, y, zl);
        if ((tm.w[1][1] >> t.lane) & 1ull) {
            // lbm_solver.cu:258-267: rho = 0, u = 0.  The reference also copies f_in -> f_out
            // here; we keep solid cells frozen in both buffers instead (k_sync_solid).
            if (MODE == 1 || ((MODE == 0 || MODE == 3) && want_moments)) {
                const long long ridx = ((long long)(zl - 1 - g.fz0) * g.H + y) * g.W + x;
                rho_out[ridx] = 0.0f;
                u_out[ridx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            if (MODE == 2 || MODE == 3) {
#pragma unroll
                for (int k = 0; k < Q; k++) f[k] = skip_solid ? weight(k) : fin[(long long)k * g.plane + own];
            }
            if (MODE == 2 || MODE == 3) {  // the neighbour's outlet branch may read a solid cell: send its frozen values
                if (zl == 1 && msg_bottom) emit_halo(g, msg_bottom, 0, x, y, f);
                if (zl == g.D && msg_top) emit_halo(g, msg_top, 1, x, y, f);
            }
            if (MODE == 3 && fout && (!skip_solid || x == g.W - 1)) {  // the lattice moves: a solid cell's frozen populations move with it
#pragma unroll
                for (int k = 0; k < Q; k++) fout[(long long)k * g.plane + own] = f[k];
            }
            return;
        }
        GatherAll<PRESET, Q - 1>::template general<(MODE == 3)>(g, fin, tm, t.lane, x, y, zl, own, inlet_e, p.U, ob_old, f);
    }

    const Moments m = moments_of(f);
    if (MODE == 1 || ((MODE == 0 || MODE == 3) && want_moments)) {
        const long long ridx = ((long long)(zl - 1 - g.fz0) * g.H + y) * g.W + x;
        if (nt_fields) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(m.rho, rho_out + ridx);
            __builtin_nontemporal_store((v4f){m.ux, m.uy, m.uz, 0.0f}, (v4f*)(u_out + ridx));
        } else {
            rho_out[ridx] = m.rho;
            u_out[ridx] = make_float4(m.ux, m.uy, m.uz, 0.0f);
        }
        if (MODE == 1) return;
    }
    float o[Q];
    collide_cell(f, m, p.omega_shear, p.omega_bulk, o);
    if (MODE != 3 || fout) {  // MODE 3 without fout: message-only pass over a slab-boundary layer
        float* dst = fout + own;
#pragma unroll
        for (int k = 0; k < Q; k++) stf<TUNE>(dst + (long long)k * g.plane, o[k]);
    }
    if (MODE == 2 || MODE == 3) {
        if (zl == 1 && msg_bottom) emit_halo(g, msg_bottom, 0, x, y, o);
        if (zl == g.D && msg_top) emit_halo(g, msg_top, 1, x, y, o);
    }
    if (MODE == 3 && !g.slab && x == g.W - 1) {
#pragma unroll
        for (int i = 0; i < OB_N; i++) ob_new[ob_at(g, i, zl - 1, y)] = o[OB_PLANE[i]];
    }
}
