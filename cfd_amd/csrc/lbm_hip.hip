// lbm_hip.hip -- MI355X (gfx950) D3Q19 lattice-Boltzmann hot path behind include/lbm_abi.h.
//
// Replaces Engine/src/Physics/lbm_solver.cu of the reference (kernels :37-64, :242-402,
// :498-513, :612-656 and the host methods around them).  Written for CDNA4 from scratch:
//   * populations: SoA fp32, 19 planes; a plane is (D+2) z-layers (one ghost layer each side for
//     the slab halo) of H rows of Wp floats, Wp = W rounded up to 32 floats (128-byte rows);
//     consecutive planes are separated by a skew so the 19+19 streams do not alias in HBM;
//   * solid mask: bit-packed, one 64-bit word per 64 cells along x (= one wave64 row segment),
//     with the same ghost layers;
//   * one wave64 owns 64 consecutive x cells of one (y,z) row.  A byte per such tile, rebuilt
//     when the mask changes, says whether the tile can take the branch-free interior path
//     (no wall row, no solid cell in the 3x3x66 neighbourhood) or needs the full boundary
//     chain of the reference (lbm_solver.cu:283-318), evaluated per lane in logical
//     (plane,x,y,z) coordinates;
//   * pull-scheme A/B: gather 19 loads -> moments -> collision -> 19 non-temporal stores,
//     152 algorithmic bytes per cell update; rho/u/curl are materialised on demand from the
//     previous buffer with the same gather, so they are bit-identical to per-step writes;
//   * arithmetic is the reference's fp32 operation order with contraction off (d3q19.h), so
//     results match the CPU oracle bit for bit.
// HBM-bound integer/float streaming work: no MFMA, no LDS staging.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lbm_abi.h"
#include "d3q19.h"
#include "voxelizer.h"

using namespace d3q19;

namespace {

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
// 6-7 % faster from HBM than 19+19 far-apart planes (profiles/r01_tuning.md, tools/membench5/6).

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
// (+2 % at 256^3, tools/tune_bench.py).
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

// full boundary chain for one direction, per lane: the branch is turned into an address by
// selects, so every lane issues exactly one load and no load waits for another
template <int PRESET, int K>
__device__ __forceinline__ float gather_general(const Geo& g, const float* __restrict__ fin, const TileMask& t,
                                                int lane, int x, int y, int zl, long long own, float inlet_e, float U) {
    const unsigned code = classify_with<PRESET, K>(g, x, y, zl, [&] { return tile_nb_solid<K>(t, lane); });
    // own-cell sources: ceiling / side mirror, bounce-back (floor, solid, wall, lid); an inlet lane
    // loads its own slot and discards it
    const int pl = code == SRC_CEIL_RFY ? RFY[K] : (code == SRC_SIDE_RFZ ? RFZ[K] : (code == SRC_INLET ? K : INV[K]));
    long long a = (long long)pl * g.plane + own;
    if (code == SRC_PULL) a = (long long)K * g.plane + own - ((long long)CZ[K] * g.layer + (long long)CY[K] * g.row + CX[K]);
    if (PRESET == LBM_PRESET_WIND_TUNNEL_REF && CX[K] == -1) {
        const long long ao = outlet_off<K>(g, y, zl);  // wave-uniform
        if (code == SRC_OUTLET) a = ao;
    }
    float v = fin[a];
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
    static __device__ __forceinline__ void general(const Geo& g, const float* __restrict__ fin, const TileMask& t,
                                                   int lane, int x, int y, int zl, long long own, float inlet_e,
                                                   float U, float* f) {
        GatherAll<PRESET, K - 1>::general(g, fin, t, lane, x, y, zl, own, inlet_e, U, f);
        f[K] = gather_general<PRESET, K>(g, fin, t, lane, x, y, zl, own, inlet_e, U);
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
    static __device__ __forceinline__ void general(const Geo&, const float*, const TileMask&, int, int, int, int,
                                                   long long, float, float, float*) {}
    static __device__ __forceinline__ void classify(const Geo&, const unsigned long long*, int, int, int,
                                                    unsigned char*) {}
};

// Tile -> (xs, y, z) mapping shared by the kernels that walk tiles: a 256-thread block is four
// waves; `segsx` of them sit side by side in x, the rest stack in y.
struct TileId {
    int xs, y, lane;
};
__device__ __forceinline__ TileId tile_of_wave(int segsx_log2) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TileId t;
    t.lane = threadIdx.x & 63;
    const int segsx = 1 << segsx_log2;
    t.xs = blockIdx.x * segsx + (wave & (segsx - 1));
    t.y = blockIdx.y * (4 >> segsx_log2) + (wave >> segsx_log2);
    return t;
}

// ------------------------------------------------------------------------------------------
// the hot kernel: fused pull-stream + boundary chain + collision (lbm_solver.cu:242-402)
// MODE 0: step (writes f_out).  MODE 1: moments only (writes rho/u in the reference layout).
// MODE 2: step of the one or two slab-boundary layers (blockIdx.z = 0 -> zl_begin, 1 -> zl_second)
//         that also writes the outgoing halo messages, so no separate pack pass is needed.
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
    int zl_second, float* __restrict__ msg_bottom, float* __restrict__ msg_top) {
    const TileId t = tile_of_wave(segsx_log2);
    const int y = t.y;
    // local layer incl. ghost offset (owned: 1..D); zrev walks the layers top-down so that a step
    // starts on the layers the previous step wrote last (still in the Infinity Cache)
    const int zl = MODE == 2 ? (blockIdx.z == 0 ? zl_begin : zl_second)
                             : zl_begin + (zrev ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z);
    if (t.xs >= g.Wt || y >= g.H) return;
    const int x = t.xs * 64 + t.lane;
    const unsigned char flag = flags[((long long)(zl - 1) * g.H + y) * g.Wt + t.xs];
    if (x >= g.W) return;
    if (MODE == 0 && flag == 2) return;  // every cell of the tile is solid: nothing moves

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
            if (MODE == 1) {
                const long long ridx = ((long long)(zl - 1 - g.fz0) * g.H + y) * g.W + x;
                rho_out[ridx] = 0.0f;
                u_out[ridx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            if (MODE == 2) {  // the neighbour's outlet branch may read a solid cell: send its frozen values
                for (int k = 0; k < Q; k++) f[k] = fin[(long long)k * g.plane + own];
                if (zl == 1 && msg_bottom) emit_halo(g, msg_bottom, 0, x, y, f);
                if (zl == g.D && msg_top) emit_halo(g, msg_top, 1, x, y, f);
            }
            return;
        }
        GatherAll<PRESET, Q - 1>::general(g, fin, tm, t.lane, x, y, zl, own, inlet_e, p.U, f);
    }

    const Moments m = moments_of(f);
    if (MODE == 1) {
        const long long ridx = ((long long)(zl - 1 - g.fz0) * g.H + y) * g.W + x;
        rho_out[ridx] = m.rho;
        u_out[ridx] = make_float4(m.ux, m.uy, m.uz, 0.0f);
        return;
    }
    float o[Q];
    collide_cell(f, m, p.omega_shear, p.omega_bulk, o);
    float* dst = fout + own;
#pragma unroll
    for (int k = 0; k < Q; k++) stf<TUNE>(dst + (long long)k * g.plane, o[k]);
    if (MODE == 2) {
        if (zl == 1 && msg_bottom) emit_halo(g, msg_bottom, 0, x, y, o);
        if (zl == g.D && msg_top) emit_halo(g, msg_top, 1, x, y, o);
    }
}

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
constexpr int OB_N = 7;
constexpr int OB_PLANE[OB_N] = {2, 8, 9, 10, 12, 13, 14};
__host__ __device__ constexpr int ob_index(int pl) { return pl == 2 ? 0 : (pl <= 10 ? pl - 7 : pl - 8); }
// side-array slot of plane-slot i, local layer z in [-1, D] (ghost rows for the z-slab halo), row y
__device__ __forceinline__ long long ob_at(const Geo& g, int i, int z, int y) {
    return ((long long)i * (g.D + 2) + (z + 1)) * g.H + y;
}

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
    if (PRESET == LBM_PRESET_WIND_TUNNEL_REF) {
        if (cx == 1) {
            const float v = ldf<TUNE>(p);  // x = 0 reads a ghost slot: in bounds, discarded
            return x == 0 ? weight(K) * inlet_e : v;
        }
        if (x == g.W - 1)  // interior tile: ny, nz are in range
            return ob_old[ob_at(g, ob_index(K), zl - 1 - CZ[K], y - CY[K])];
        return ldf<TUNE>(p);
    } else {
        const bool wall = (cx == 1) ? (x == 0) : (x == g.W - 1);
        if (wall) return *aa_own_pop<INV[K], PHASE>(g, M, own);
        return ldf<TUNE>(p);
    }
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
        stf<TUNE>(dst, o[K]);
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
    const TileId t = tile_of_wave(segsx_log2);
    const int y = t.y;
    // zl_second > 0: the two slab-boundary layers (blockIdx.z = 0 -> zl_begin, 1 -> zl_second)
    const int zl = zl_second > 0 ? (blockIdx.z == 0 ? zl_begin : zl_second)
                                 : zl_begin + (zrev ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z);
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

// ------------------------------------------------------------------------------------------
// set-up and read-back kernels
// ------------------------------------------------------------------------------------------
// k_init (lbm_solver.cu:37-64): equilibrium at rho = 1, u = 0 is f_k = w_k exactly.  Fills the
// whole plane (ghost layers and row padding too) of both buffers.
// Every float of the allocation [first, last) relative to plane 0 is filled (ghost layers, row
// padding, skew and the AA front guard too), with the weight of the population whose run it
// belongs to.
__global__ void k_fill_planes(float* __restrict__ fa, float* __restrict__ fb, Geo g, int rows_layout,
                              long long first, long long last) {
    const long long i = first + gidx();
    if (i >= last) return;
    long long k;
    if (rows_layout) {
        long long r = i % g.row;
        if (r < 0) r += g.row;
        k = r / g.Wp;
    } else {
        k = i < 0 ? 0 : i / g.plane;
    }
    const float w = weight((int)(k < Q ? k : Q - 1));
    fa[i] = w;
    fb[i] = w;
}

__global__ void k_fill_fields(float* __restrict__ rho, float4* __restrict__ u, long long n) {
    const long long i = gidx();
    if (i >= n) return;
    rho[i] = 1.0f;
    u[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// dense u8 mask (reference wire format) -> bit-packed words of the owned layers
__global__ void k_pack_mask(const unsigned char* __restrict__ dense, unsigned long long* __restrict__ mask, Geo g) {
    const long long wi = gidx();  // word index over owned rows
    const long long nwords = (long long)g.D * g.H * g.Wm;
    if (wi >= nwords) return;
    const long long row = wi / g.Wm;
    const int xw = (int)(wi - row * g.Wm);
    const unsigned char* src = dense + row * g.W + (long long)xw * 64;
    const int cnt = min(64, g.W - xw * 64);
    unsigned long long bits = 0;
    for (int b = 0; b < cnt; b++) bits |= (unsigned long long)(src[b] > 0) << b;
    mask[(long long)g.H * g.Wm + wi] = bits;  // skip the bottom ghost layer
}

__global__ void k_unpack_mask(const unsigned long long* __restrict__ mask, unsigned char* __restrict__ dense, Geo g) {
    const long long i = gidx();
    const long long n = (long long)g.W * g.H * g.D;
    if (i >= n) return;
    const long long row = i / g.W;
    const int x = (int)(i - row * g.W);
    const unsigned long long w = mask[(long long)g.H * g.Wm + row * g.Wm + (x >> 6)];
    dense[i] = ((w >> (x & 63)) & 1ull) ? 255 : 0;
}

// k_add_solid_rect (lbm_solver.cu:498-513) on the packed mask: one thread per word, no atomics.
// Coordinates are global; this context owns global layers [z0, z0+D).
__global__ void k_add_rect(unsigned long long* __restrict__ mask, Geo g, int rx, int ry, int rz, int rw, int rh, int rd) {
    const long long wi = gidx();
    const long long nwords = (long long)(g.D + 2) * g.H * g.Wm;  // ghost layers included
    if (wi >= nwords) return;
    const long long row = wi / g.Wm;
    const int xw = (int)(wi - row * g.Wm);
    const int zl = (int)(row / g.H);
    const int y = (int)(row - (long long)zl * g.H);
    const int zg = zl - 1 + g.z0;
    if (zg < 0 || zg >= g.Dg) return;
    // 64-bit compares: rx + rw may exceed int range for hostile arguments
    if (y < ry || (long long)y >= (long long)ry + rh || zg < rz || (long long)zg >= (long long)rz + rd) return;
    long long lo = max((long long)rx, (long long)xw * 64);
    long long hi = min(min((long long)rx + rw, (long long)g.W), (long long)xw * 64 + 64);
    if (hi <= lo) return;
    const int b0 = (int)(lo - (long long)xw * 64), cnt = (int)(hi - lo);
    const unsigned long long bits = (cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull)) << b0;
    mask[wi] |= bits;
}

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
    if ((mask[((long long)(z + 1) * g.H + y) * g.Wm + xs] & all) == all) {
        fl = 2;
    } else if (y == 0 || y == g.H - 1 || zg == 0 || zg == g.Dg - 1) {
        fl = 1;
    } else {
        unsigned long long any = 0;
        for (int dz = -1; dz <= 1; dz++)
            for (int dy = -1; dy <= 1; dy++) {
                const unsigned long long* r = mask + ((long long)(z + 1 + dz) * g.H + (y + dy)) * g.Wm;
                any |= r[xs];
                if (xs > 0) any |= r[xs - 1] >> 63;
                if (xs + 1 < g.Wm) any |= r[xs + 1] & 1ull;
            }
        fl = any != 0;
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

// ---- z-slab halo messages -------------------------------------------------------------------
// message = 5 population layers (layer floats each) followed by `tail` floats of outlet-edge values:
//   top message   (side 1, from layer zl = D): planes 5,11,14,15,18; tail = f10(W-1,H-1), f9(W-1,H-1), f13(W-1,y) for all y
//   bottom message(side 0, from layer zl = 1): planes 6,12,13,16,17; tail = f8(W-1,0),  f9(W-1,0)
// The tails are what the unchecked outlet index of the reference (lbm_solver.cu:292-296) reads
// across a z face; they land in ghost-layer planes that the regular exchange does not use.
__device__ __forceinline__ int halo_plane(int side, int j) { return side ? UP5[j] : DOWN5[j]; }

__global__ void k_halo_pack(const float* __restrict__ f, float* __restrict__ msg, Geo g, int side) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    const int zl = side ? g.D : 1;
    if (j < 5) {
        if (i < g.mlayer) msg[(long long)j * g.mlayer + i] = f[(long long)halo_plane(side, j) * g.plane + cell_off(g, (int)(i % g.Wp), (int)(i / g.Wp), zl)];
        return;
    }
    float* tail = msg + 5 * g.mlayer;
    if (i >= g.H + 2) return;
    float v = 0.0f;
    if (side) {
        if (i == 0) v = f[10 * g.plane + cell_off(g, g.W - 1, g.H - 1, zl)];
        else if (i == 1) v = f[9 * g.plane + cell_off(g, g.W - 1, g.H - 1, zl)];
        else v = f[13 * g.plane + cell_off(g, g.W - 1, (int)i - 2, zl)];
    } else {
        if (i == 0) v = f[8 * g.plane + cell_off(g, g.W - 1, 0, zl)];
        else if (i == 1) v = f[9 * g.plane + cell_off(g, g.W - 1, 0, zl)];
    }
    tail[i] = v;
}

// side 0: message from the lower neighbour's TOP face -> bottom ghost layer (zl = 0)
// side 1: message from the upper neighbour's BOTTOM face -> top ghost layer (zl = D+1)
__global__ void k_halo_unpack(float* __restrict__ f, const float* __restrict__ msg, Geo g, int side) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    const int zl = side ? g.D + 1 : 0;
    if (j < 5) {
        // the message that arrives on our bottom face is the sender's top message (planes UP5)
        if (i < g.mlayer) f[(long long)halo_plane(1 - side, j) * g.plane + cell_off(g, (int)(i % g.Wp), (int)(i / g.Wp), zl)] = msg[(long long)j * g.mlayer + i];
        return;
    }
    const float* tail = msg + 5 * g.mlayer;
    if (side == 0) {
        if (i >= g.H + 2) return;
        if (i == 0) f[10 * g.plane + cell_off(g, g.W - 1, g.H - 1, zl)] = tail[0];
        else if (i == 1) f[9 * g.plane + cell_off(g, g.W - 1, g.H - 1, zl)] = tail[1];
        else f[13 * g.plane + cell_off(g, g.W - 1, (int)i - 2, zl)] = tail[i];
    } else {
        if (i == 0) f[8 * g.plane + cell_off(g, g.W - 1, 0, zl)] = tail[0];
        else if (i == 1) f[9 * g.plane + cell_off(g, g.W - 1, 0, zl)] = tail[1];
    }
}

__global__ void k_halo_unpack2(float* __restrict__ f, const float* __restrict__ from_below,
                               const float* __restrict__ from_above, Geo g) {
    const int side = blockIdx.y / 6;
    const float* msg = side ? from_above : from_below;
    if (!msg) return;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y % 6;
    const int zl = side ? g.D + 1 : 0;
    if (j < 5) {
        if (i < g.mlayer) f[(long long)halo_plane(1 - side, j) * g.plane + cell_off(g, (int)(i % g.Wp), (int)(i / g.Wp), zl)] = msg[(long long)j * g.mlayer + i];
        return;
    }
    const float* tail = msg + 5 * g.mlayer;
    if (side == 0) {
        if (i >= g.H + 2) return;
        if (i == 0) f[10 * g.plane + cell_off(g, g.W - 1, g.H - 1, zl)] = tail[0];
        else if (i == 1) f[9 * g.plane + cell_off(g, g.W - 1, g.H - 1, zl)] = tail[1];
        else f[13 * g.plane + cell_off(g, g.W - 1, (int)i - 2, zl)] = tail[i];
    } else {
        if (i == 0) f[8 * g.plane + cell_off(g, g.W - 1, 0, zl)] = tail[0];
        else if (i == 1) f[9 * g.plane + cell_off(g, g.W - 1, 0, zl)] = tail[1];
    }
}

// ---- AA z-slab halo ---------------------------------------------------------------------------
// Which slots carry the five populations that cross a face depends on the layout of the moment:
//   swapped  (after an EVEN step): F_k of a boundary-layer cell sits in the OWNED layer at plane
//            inv(k); it goes to the same plane of the neighbour's GHOST layer, where the
//            neighbour's odd step pulls it from;
//   streamed (after an ODD step, or after init / import): the cell has pushed F_k into OUR GHOST
//            layer at plane k; it goes to the neighbour's OWNED boundary layer, same plane, where
//            the neighbour's even step reads its own slots.
// Only cells x < W, 0 <= y < H travel: the x/y ghost shell of a layer belongs to the wall logic of
// the slab that wrote it.  Message = 5 layers of H*Wp floats + the 7 outlet-column planes of the
// boundary layer (7*H floats, what aa_outlet_read needs across the face, ring link included).
__global__ void k_aa_halo_pack(const float* __restrict__ M, const float* __restrict__ ob, float* __restrict__ msg,
                               Geo g, int side, int streamed) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (j < 5) {
        if (i >= g.mlayer) return;
        const int x = (int)(i % g.Wp), y = (int)(i / g.Wp);
        if (x >= g.W) return;
        const int zl = streamed ? (side ? g.D + 1 : 0) : (side ? g.D : 1);
        const int pl = streamed ? (side ? UP5[j] : DOWN5[j]) : (side ? DOWN5[j] : UP5[j]);
        msg[(long long)j * g.mlayer + i] = M[(long long)pl * g.plane + cell_off(g, x, y, zl)];
        return;
    }
    if (i >= (long long)OB_N * g.H) return;
    msg[5 * g.mlayer + i] = ob[ob_at(g, (int)(i / g.H), side ? g.D - 1 : 0, (int)(i % g.H))];
}
// side: the face of THIS slab the message arrived on (0: from the lower neighbour's top face)
__global__ void k_aa_halo_unpack(float* __restrict__ M, float* __restrict__ ob, const float* __restrict__ msg,
                                 Geo g, int side, int streamed) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (j < 5) {
        if (i >= g.mlayer) return;
        // the ring link of the wind-tunnel preset carries only the outlet column: beyond a domain end
        // there is no cell, and the ghost slots there belong to this slab's own wall logic
        if (side ? (g.z0 + g.D == g.Dg) : (g.z0 == 0)) return;
        const int x = (int)(i % g.Wp), y = (int)(i / g.Wp);
        if (x >= g.W) return;
        const int zl = streamed ? (side ? g.D : 1) : (side ? g.D + 1 : 0);
        const int pl = streamed ? (side ? DOWN5[j] : UP5[j]) : (side ? UP5[j] : DOWN5[j]);
        M[(long long)pl * g.plane + cell_off(g, x, y, zl)] = msg[(long long)j * g.mlayer + i];
        return;
    }
    if (i >= (long long)OB_N * g.H) return;
    ob[ob_at(g, (int)(i / g.H), side ? g.D : -1, (int)(i % g.H))] = msg[5 * g.mlayer + i];
}

// one layer of the packed mask <-> dense u8 (ghost exchange of the static mask)
__global__ void k_mask_layer_get(const unsigned long long* __restrict__ mask, unsigned char* __restrict__ dense, Geo g, int zl) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)g.W * g.H) return;
    const int y = (int)(i / g.W), x = (int)(i - (long long)y * g.W);
    dense[i] = solid_at(g, mask, x, y, zl) ? 255 : 0;
}
__global__ void k_mask_layer_set(unsigned long long* __restrict__ mask, const unsigned char* __restrict__ dense, Geo g, int zl) {
    const long long wi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= (long long)g.H * g.Wm) return;
    const int y = (int)(wi / g.Wm), xw = (int)(wi - (long long)y * g.Wm);
    const unsigned char* src = dense + (long long)y * g.W + (long long)xw * 64;
    const int cnt = min(64, g.W - xw * 64);
    unsigned long long bits = 0;
    for (int b = 0; b < cnt; b++) bits |= (unsigned long long)(src[b] > 0) << b;
    mask[(long long)zl * g.H * g.Wm + wi] = bits;
}

// k_compute_curl (lbm_solver.cu:612-656): |curl u| by clamped central differences.
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

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
enum FieldState { FIELDS_INIT = 0, FIELDS_STALE = 1, FIELDS_VALID = 2 };

struct lbm_ctx {
    Geo g{};
    int device = 0;
    int preset = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream_b = nullptr;  // optional caller-owned stream for the slab-boundary layers
    bool own_stream = false;
    float* f[2] = {nullptr, nullptr};
    int cur = 0;
    unsigned long long* mask = nullptr;
    unsigned char* flags = nullptr;
    float* rho = nullptr;
    float4* u = nullptr;
    float* curl = nullptr;
    unsigned char* solid_dense = nullptr;
    Params p{0.08f, 1.98f, 1.90f};  // lbm_solver.cuh:11-13
    Params p_last{0.08f, 1.98f, 1.90f};
    bool mask_dirty = true;    // tile flags / solid sync pending
    FieldState fields = FIELDS_INIT;
    bool curl_valid = false;
    int segsx_log2 = 0;
    int split_phase = 0;
    int tune = 0;              // bit 0 nt loads, bit 1 sc1 stores, bit 2 alternate z direction per step
    int zrev = 0;
    int scheme = LBM_SCHEME_AB;
    int aa_phase = 0;          // AA: 0 = next step is even (buffer in the streamed layout)
    float* alloc0 = nullptr;   // AA: start of the allocation (f[0] sits behind a front guard)
    float* ob[2] = {nullptr, nullptr};
    int ob_cur = 0;
    unsigned long long* prev_mask = nullptr;
    size_t mask_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint64_t steps = 0, steps_at_begin = 0;
    double last_ms = 0.0;
    uint64_t last_steps = 0;
    size_t pop_bytes = 0;
    long long pop_floats = 0;  // floats of one population buffer behind f[i]
    long long guard_floats = 0;  // AA: floats allocated in front of f[0]
    int layout = LBM_LAYOUT_ROWS;
    unsigned lds_pad = 0;      // dynamic LDS bytes per block of the step kernel: caps the blocks per CU (occupancy throttle)
    std::string err;
};

static thread_local std::string g_create_error;

static int fail(lbm_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((c), LBM_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                  \
    } while (0)

#define NEED(c)                                               \
    do {                                                      \
        if (!(c)) return fail(nullptr, LBM_ERR_INVALID, "null context"); \
        hipError_t e_ = hipSetDevice((c)->device);            \
        if (e_ != hipSuccess) return fail((c), LBM_ERR_NO_DEVICE, "hipSetDevice(%d): %s", (c)->device, hipGetErrorString(e_)); \
    } while (0)

static inline long long cells(const lbm_ctx* c) { return (long long)c->g.W * c->g.H * c->g.D; }
static inline long long field_cells(const lbm_ctx* c) { return (long long)c->g.W * c->g.H * c->g.fnz; }
static inline unsigned blocks_for(long long n, int bs) { return (unsigned)((n + bs - 1) / bs); }
// grid of the 1-D kernels that index with gidx(): at most 2^20 blocks in x, the rest in y
static inline dim3 grid1(long long n, int bs = 256) {
    const long long blocks = (n + bs - 1) / bs;
    const long long gx = blocks < (1ll << 20) ? (blocks > 0 ? blocks : 1) : (1ll << 20);
    return dim3((unsigned)gx, (unsigned)((blocks + gx - 1) / gx));
}

static int ensure_field_buffers(lbm_ctx* c) {
    const long long n = field_cells(c);
    if (n == 0) return LBM_OK;
    if (!c->rho) HIPCHK(c, hipMalloc(&c->rho, n * sizeof(float)));
    if (!c->u) HIPCHK(c, hipMalloc(&c->u, n * sizeof(float4)));
    return LBM_OK;
}

// rebuild tile flags and re-freeze solid cells after a mask / population change
static int refresh_mask_state(lbm_ctx* c) {
    if (!c->mask_dirty) return LBM_OK;
    const long long ntiles = (long long)c->g.D * c->g.H * c->g.Wt;
    if (c->preset == LBM_PRESET_CAVITY)
        hipLaunchKernelGGL(k_tile_flags<LBM_PRESET_CAVITY>, grid1(ntiles), dim3(256), 0, c->stream, c->mask, c->flags, c->g);
    else
        hipLaunchKernelGGL(k_tile_flags<LBM_PRESET_WIND_TUNNEL_REF>, grid1(ntiles), dim3(256), 0, c->stream, c->mask, c->flags, c->g);
    HIPCHK(c, hipGetLastError());
    if (c->scheme == LBM_SCHEME_AA) {
        hipLaunchKernelGGL(k_aa_refluidize, grid1(cells(c)), dim3(256), 0, c->stream, c->f[0],
                           c->ob[c->ob_cur], c->prev_mask, c->mask, c->g, c->aa_phase);
        HIPCHK(c, hipGetLastError());
        hipLaunchKernelGGL(k_ob_sync, grid1((long long)c->g.D * c->g.H), dim3(256), 0, c->stream,
                           c->ob[c->ob_cur], c->ob[1 - c->ob_cur], c->mask, c->g);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(c->prev_mask, c->mask, c->mask_bytes, hipMemcpyDeviceToDevice, c->stream));
    } else {
        hipLaunchKernelGGL(k_sync_solid, grid1(cells(c)), dim3(256), 0, c->stream,
                           c->f[c->cur], c->f[1 - c->cur], c->mask, c->g);
        HIPCHK(c, hipGetLastError());
    }
    c->mask_dirty = false;
    return LBM_OK;
}

template <int MODE>
static int launch_stream_collide(lbm_ctx* c, const float* fin, float* fout, const Params& p, int zl_begin, int nlayers,
                                 int zl_second = 0, float* msg_bottom = nullptr, float* msg_top = nullptr) {
    if (nlayers <= 0) return LBM_OK;
    hipStream_t stream = (MODE == 2 && c->stream_b) ? c->stream_b : c->stream;
    const int segsx = 1 << c->segsx_log2;
    dim3 grid((c->g.Wt + segsx - 1) / segsx, (c->g.H + (4 / segsx) - 1) / (4 / segsx), nlayers);
    const int zrev = (MODE == 0 && (c->tune & 4)) ? c->zrev : 0;
#define LBM_LAUNCH(PRESET, TUNE)                                                                              \
    hipLaunchKernelGGL((k_stream_collide<PRESET, MODE, TUNE>), grid, dim3(256), c->lds_pad, stream, fin, fout, c->mask, \
                       c->flags, c->rho, c->u, c->g, p, zl_begin, c->segsx_log2, zrev, zl_second, msg_bottom, msg_top)
#define LBM_LAUNCH_T(PRESET)                          \
    switch (c->tune & 3) {                            \
        case 0: LBM_LAUNCH(PRESET, 0); break;         \
        case 1: LBM_LAUNCH(PRESET, 1); break;         \
        case 2: LBM_LAUNCH(PRESET, 2); break;         \
        default: LBM_LAUNCH(PRESET, 3); break;        \
    }
    if (c->preset == LBM_PRESET_CAVITY) { LBM_LAUNCH_T(LBM_PRESET_CAVITY) }
    else { LBM_LAUNCH_T(LBM_PRESET_WIND_TUNNEL_REF) }
#undef LBM_LAUNCH_T
#undef LBM_LAUNCH
    HIPCHK(c, hipGetLastError());
    return LBM_OK;
}

// one AA launch over layers [zl_begin, zl_begin + nlayers), or over the two layers zl_begin and
// zl_second when zl_second > 0; the caller flips the phase with aa_flip() once the step is complete
static int launch_aa(lbm_ctx* c, int want_moments, int zl_begin, int nlayers, int zl_second, hipStream_t stream = nullptr) {
    if (nlayers <= 0) return LBM_OK;
    if (!stream) stream = c->stream;
    const int segsx = 1 << c->segsx_log2;
    dim3 grid((c->g.Wt + segsx - 1) / segsx, (c->g.H + (4 / segsx) - 1) / (4 / segsx), nlayers);
    const int zrev = ((c->tune & 4) && nlayers == c->g.D) ? c->zrev : 0;
#define LBM_AA(PRESET, PHASE, TUNE)                                                                               \
    hipLaunchKernelGGL((k_aa_step<PRESET, PHASE, TUNE>), grid, dim3(256), 0, stream, c->f[0], c->ob[c->ob_cur],     \
                       c->ob[1 - c->ob_cur], c->mask, c->flags, c->rho, c->u, c->g, c->p, c->segsx_log2, zrev,     \
                       want_moments, zl_begin, zl_second)
#define LBM_AA_T(PRESET, PHASE)                       \
    switch (c->tune & 3) {                            \
        case 0: LBM_AA(PRESET, PHASE, 0); break;      \
        case 1: LBM_AA(PRESET, PHASE, 1); break;      \
        case 2: LBM_AA(PRESET, PHASE, 2); break;      \
        default: LBM_AA(PRESET, PHASE, 3); break;     \
    }
#define LBM_AA_P(PRESET) \
    if (c->aa_phase == 0) { LBM_AA_T(PRESET, 0) } else { LBM_AA_T(PRESET, 1) }
    if (c->preset == LBM_PRESET_CAVITY) { LBM_AA_P(LBM_PRESET_CAVITY) }
    else { LBM_AA_P(LBM_PRESET_WIND_TUNNEL_REF) }
#undef LBM_AA_P
#undef LBM_AA_T
#undef LBM_AA
    HIPCHK(c, hipGetLastError());
    return LBM_OK;
}
static void aa_flip(lbm_ctx* c) {
    c->aa_phase ^= 1;
    c->ob_cur ^= 1;
    c->zrev ^= 1;
}
static int launch_aa_step(lbm_ctx* c, int want_moments) {
    int rc = launch_aa(c, want_moments, 1, c->g.D, 0);
    if (!rc) aa_flip(c);
    return rc;
}

// bring rho/u up to date: what the reference's per-step writes would hold right now
static int materialise_fields(lbm_ctx* c) {
    if (c->g.fnz == 0) return fail(c, LBM_ERR_STATE, "the field window is empty (lbm_set_field_window)");
    if (c->fields == FIELDS_VALID) return LBM_OK;
    int rc = ensure_field_buffers(c);
    if (rc) return rc;
    if (c->fields == FIELDS_INIT) {
        hipLaunchKernelGGL(k_fill_fields, grid1(field_cells(c)), dim3(256), 0, c->stream, c->rho, c->u, field_cells(c));
        HIPCHK(c, hipGetLastError());
    } else if (c->scheme == LBM_SCHEME_AA) {
        return fail(c, LBM_ERR_STATE, "AA scheme: rho/u of a new field window exist after the next lbm_step()");
    } else {
        // the last step read f[1-cur]; it is still intact (A/B), so the same gather gives the
        // same post-stream moments (lbm_solver.cu:320-337)
        rc = launch_stream_collide<1>(c, c->f[1 - c->cur], nullptr, c->p_last, 1 + c->g.fz0, c->g.fnz);
        if (rc) return rc;
    }
    c->fields = FIELDS_VALID;
    c->curl_valid = false;
    return LBM_OK;
}

extern "C" {

int lbm_abi_version(void) { return LBM_ABI_VERSION; }

const char* lbm_last_error(lbm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int lbm_create(lbm_ctx** out, int width, int height, int depth, const lbm_opts* opts) {
    if (!out) return fail(nullptr, LBM_ERR_INVALID, "lbm_create: out is null");
    *out = nullptr;
    if (width < 1 || height < 1 || depth < 1)
        return fail(nullptr, LBM_ERR_INVALID, "lbm_create: dimensions must be positive (got %d x %d x %d)", width, height, depth);
    lbm_opts o;
    memset(&o, 0, sizeof o);
    o.device = -1;
    o.plane_skew_bytes = -1;
    if (opts) {
        if (opts->struct_size < (int)offsetof(lbm_opts, reserved) || opts->struct_size > (int)sizeof(lbm_opts))
            return fail(nullptr, LBM_ERR_INVALID, "lbm_create: lbm_opts.struct_size %d not understood", opts->struct_size);
        memcpy(&o, opts, (size_t)opts->struct_size);
    }
    if (o.preset != LBM_PRESET_WIND_TUNNEL_REF && o.preset != LBM_PRESET_CAVITY)
        return fail(nullptr, LBM_ERR_INVALID, "lbm_create: unknown preset %d", o.preset);
    if (o.scheme != LBM_SCHEME_AB && o.scheme != LBM_SCHEME_AA)
        return fail(nullptr, LBM_ERR_INVALID, "lbm_create: unknown scheme %d", o.scheme);
    int Dg = o.depth_global > 0 ? o.depth_global : depth;
    int z0 = o.depth_global > 0 ? o.z0 : 0;
    if (z0 < 0 || (long long)z0 + depth > Dg)
        return fail(nullptr, LBM_ERR_INVALID, "lbm_create: slab [%d, %d) outside global depth %d", z0, z0 + depth, Dg);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, LBM_ERR_NO_DEVICE, "lbm_create: no HIP device available (this library has no CPU path)");
    int dev = o.device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev >= ndev) return fail(nullptr, LBM_ERR_NO_DEVICE, "lbm_create: device %d of %d", dev, ndev);
    if (hipSetDevice(dev) != hipSuccess) return fail(nullptr, LBM_ERR_NO_DEVICE, "lbm_create: hipSetDevice(%d) failed", dev);

    lbm_ctx* c = new lbm_ctx();
    c->device = dev;
    c->preset = o.preset;
    Geo& g = c->g;
    g.W = width; g.H = height; g.D = depth; g.Dg = Dg; g.z0 = z0;
    g.slab = o.depth_global > 0 ? 1 : 0;
    g.fz0 = 0; g.fnz = depth;
    c->scheme = o.scheme;
    const bool aa = o.scheme == LBM_SCHEME_AA;
    int wp_align = 32;
    if (const char* t = getenv("LBM_WP_ALIGN")) wp_align = std::max(32, atoi(t) / 32 * 32);
    g.Wp = ((aa ? width + 2 : width) + wp_align - 1) / wp_align * wp_align;  // AA: ghost slots x = -1 and x = W in the row padding
    g.Wm = (width + 63) / 64;
    g.Wt = g.Wm;
    g.mlayer = (long long)height * g.Wp;
    int layout = o.layout ? o.layout : LBM_LAYOUT_ROWS;  // measured faster, see the note at struct Geo
    if (const char* t = getenv("LBM_LAYOUT")) layout = atoi(t);
    if (layout != LBM_LAYOUT_PLANES && layout != LBM_LAYOUT_ROWS) {
        delete c;
        return fail(nullptr, LBM_ERR_INVALID, "lbm_create: unknown layout %d", layout);
    }
    c->layout = layout;
    const long long rows_per_layer = aa ? height + 2 : height;  // AA: ghost rows y = H and y = -1
    long long guard;                                             // AA: floats in front of plane 0 that (y = -1, z = -1) reaches
    if (layout == LBM_LAYOUT_ROWS) {
        long long row_pad = 0;
        if (const char* t = getenv("LBM_ROW_PAD")) row_pad = atoll(t) / 32 * 32;
        g.plane = g.Wp;
        g.row = (long long)Q * g.Wp + (row_pad > 0 ? row_pad : 0);
        g.layer = rows_per_layer * g.row;
        guard = aa ? g.row + 32 : 0;
        c->pop_floats = (long long)(depth + 2) * g.layer;
    } else {
        g.row = g.Wp;
        g.layer = rows_per_layer * g.Wp;
        long long skew_floats = (o.plane_skew_bytes >= 0 ? o.plane_skew_bytes : 3072) / 4;  // measured best of a 0..1 MiB sweep
        guard = aa ? g.Wp + 32 : 0;
        if (skew_floats < guard) skew_floats = guard;
        skew_floats = (skew_floats + 31) / 32 * 32;                                          // keep rows 128-B aligned
        g.plane = (long long)(depth + 2) * g.layer + skew_floats;
        c->pop_floats = (long long)Q * g.plane;
    }
    c->guard_floats = guard;
    c->segsx_log2 = g.Wt >= 4 ? 2 : (g.Wt >= 2 ? 1 : 0);
    // measured on MI355X at 256^3 (profiles/r01_tuning.md): nt loads + sc1 (write-through) stores +
    // alternating z direction = +2.5 % over plain loads / nt stores.  LBM_TUNE overrides for A/B runs.
    // In place (AA) the write-through stores lose: partial-line neighbour writes no longer combine
    // in L2 (4.24 vs 5.39 TB/s), so AA keeps nt loads + nt stores + alternating z direction.
    // Row-interleaved A/B (tools/tune_bench.py, 256^3): nt loads + nt stores 6.08 TB/s, with sc1 stores
    // 5.5, with alternating z 6.03 -> plain nt/nt is the default there.
    c->tune = o.scheme == LBM_SCHEME_AA ? 5 : (layout == LBM_LAYOUT_ROWS ? 1 : 7);
    if (const char* t = getenv("LBM_TUNE")) c->tune = atoi(t);
    if (const char* t = getenv("LBM_LDS_PAD")) c->lds_pad = (unsigned)atoi(t);
    c->pop_bytes = (size_t)c->pop_floats * sizeof(float);

    auto cleanup = [&](int code) {
        lbm_destroy(c);
        return code;
    };
#define CR(call)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fail(nullptr, LBM_ERR_HIP, "lbm_create: %s failed: %s", #call, hipGetErrorString(e_)); \
            return cleanup(LBM_ERR_HIP);                                                           \
        }                                                                                          \
    } while (0)
    if (o.stream) {
        c->stream = (hipStream_t)o.stream;
        c->stream_b = (hipStream_t)o.boundary_stream;
    } else {
        CR(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    const size_t mask_bytes = (size_t)(depth + 2) * height * g.Wm * sizeof(unsigned long long);
    c->mask_bytes = mask_bytes;
    if (aa) {
        CR(hipMalloc(&c->alloc0, c->pop_bytes + (size_t)guard * sizeof(float)));
        c->f[0] = c->alloc0 + guard;
        const size_t ob_bytes = (size_t)OB_N * (depth + 2) * height * sizeof(float);
        CR(hipMalloc(&c->ob[0], ob_bytes));
        CR(hipMalloc(&c->ob[1], ob_bytes));
        CR(hipMalloc(&c->prev_mask, mask_bytes));
        CR(hipMemsetAsync(c->prev_mask, 0, mask_bytes, c->stream));
    } else {
        CR(hipMalloc(&c->f[0], c->pop_bytes));
        CR(hipMalloc(&c->f[1], c->pop_bytes));
    }
    CR(hipMalloc(&c->mask, mask_bytes));
    CR(hipMemsetAsync(c->mask, 0, mask_bytes, c->stream));  // d_solid.memset(0), lbm_solver.cu:415
    CR(hipMalloc(&c->flags, (size_t)depth * height * g.Wt));
    CR(hipEventCreate(&c->ev0));
    CR(hipEventCreate(&c->ev1));
#undef CR
    *out = c;
    return LBM_OK;
}

int lbm_destroy(lbm_ctx* c) {
    if (!c) return LBM_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->alloc0) hipFree(c->alloc0); else hipFree(c->f[0]);
    hipFree(c->f[1]); hipFree(c->mask); hipFree(c->flags);
    hipFree(c->ob[0]); hipFree(c->ob[1]); hipFree(c->prev_mask);
    hipFree(c->rho); hipFree(c->u); hipFree(c->curl); hipFree(c->solid_dense);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
    return LBM_OK;
}

int lbm_init(lbm_ctx* c) {
    NEED(c);
    // the whole allocation: the AA scheme's ghost slots (y = -1, z = -1) lie in the skew region behind
    // the previous plane / in the front guard
    const long long first = -c->guard_floats, last = c->pop_floats;
    hipLaunchKernelGGL(k_fill_planes, grid1(last - first), dim3(256), 0, c->stream, c->f[0],
                       c->f[1] ? c->f[1] : c->f[0], c->g, c->layout == LBM_LAYOUT_ROWS ? 1 : 0, first, last);
    HIPCHK(c, hipGetLastError());
    if (c->scheme == LBM_SCHEME_AA) {
        const long long per = (long long)(c->g.D + 2) * c->g.H;
        hipLaunchKernelGGL(k_ob_fill, grid1(per * OB_N), dim3(256), 0, c->stream, c->ob[0], c->ob[1], per);
        HIPCHK(c, hipGetLastError());
        c->aa_phase = 0;  // w_k = w_inv(k): the equilibrium fill is valid in the streamed layout
    }
    c->fields = FIELDS_INIT;  // rho = 1, u = 0 everywhere, solid cells included (lbm_solver.cu:62-63)
    c->curl_valid = false;
    return LBM_OK;
}

int lbm_set_params(lbm_ctx* c, float inlet_velocity, float omega_shear, float omega_bulk) {
    if (!c) return fail(nullptr, LBM_ERR_INVALID, "null context");
    c->p.U = inlet_velocity; c->p.omega_shear = omega_shear; c->p.omega_bulk = omega_bulk;
    return LBM_OK;
}

int lbm_get_params(lbm_ctx* c, float* inlet_velocity, float* omega_shear, float* omega_bulk) {
    if (!c) return fail(nullptr, LBM_ERR_INVALID, "null context");
    if (inlet_velocity) *inlet_velocity = c->p.U;
    if (omega_shear) *omega_shear = c->p.omega_shear;
    if (omega_bulk) *omega_bulk = c->p.omega_bulk;
    return LBM_OK;
}

// the fields the reference would hold were written with the OLD mask: settle them first
static int before_mask_change(lbm_ctx* c) {
    if (c->fields == FIELDS_STALE) {
        int rc = refresh_mask_state(c);
        if (rc) return rc;
        return materialise_fields(c);
    }
    return LBM_OK;
}

int lbm_add_solid_rect(lbm_ctx* c, int x, int y, int z, int w, int h, int d) {
    NEED(c);
    int rc = before_mask_change(c);
    if (rc) return rc;
    const long long nwords = (long long)(c->g.D + 2) * c->g.H * c->g.Wm;
    hipLaunchKernelGGL(k_add_rect, grid1(nwords), dim3(256), 0, c->stream, c->mask, c->g, x, y, z, w, h, d);
    HIPCHK(c, hipGetLastError());
    c->mask_dirty = true;
    return LBM_OK;
}

int lbm_set_solid_u8(lbm_ctx* c, const uint8_t* mask, size_t n) {
    NEED(c);
    if (!mask) return fail(c, LBM_ERR_INVALID, "lbm_set_solid_u8: mask is null");
    if (n != (size_t)cells(c)) return fail(c, LBM_ERR_INVALID, "lbm_set_solid_u8: expected %lld cells, got %zu", cells(c), n);
    int rc = before_mask_change(c);
    if (rc) return rc;
    unsigned char* dense = nullptr;
    HIPCHK(c, hipMalloc(&dense, n));
    hipError_t e = hipMemcpyAsync(dense, mask, n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const long long nwords = (long long)c->g.D * c->g.H * c->g.Wm;
        hipLaunchKernelGGL(k_pack_mask, grid1(nwords), dim3(256), 0, c->stream, dense, c->mask, c->g);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dense);
    if (e != hipSuccess) return fail(c, LBM_ERR_HIP, "lbm_set_solid_u8: %s", hipGetErrorString(e));
    c->mask_dirty = true;
    return LBM_OK;
}

int lbm_load_mesh(lbm_ctx* c, const char* obj_path) {
    NEED(c);
    if (!obj_path) return fail(c, LBM_ERR_INVALID, "lbm_load_mesh: path is null");
    const Geo& g = c->g;
    // host stage, O(triangles): parse, rotate, bounds, auto-scale (lbm_solver.cu:526-538)
    lbmvox::Mesh mesh;
    const bool loaded = mesh.load_obj(obj_path);
    if (!loaded) mesh.set_triangles(nullptr, 0);
    mesh.rotate_y(-90.0f);
    const lbmvox::Vec3 lo = mesh.min_bound(), hi = mesh.max_bound();
    const float sx = ((float)g.W * 0.8f) / (hi.x - lo.x);
    const float sy = ((float)g.H * 0.8f) / (hi.y - lo.y);
    const float sz = ((float)g.Dg * 0.8f) / (hi.z - lo.z);
    const float scale = std::min({sx, sy, sz});
    int rc = before_mask_change(c);
    if (rc) return rc;
    bool on_device = true;
    if (const char* t = getenv("LBM_VOXELIZE")) on_device = strcmp(t, "host") != 0;
    if (on_device) {
        // device stage, O(D*H*triangles): the ray casting (mesh_voxelizer.cpp:146-216), straight into the packed mask
        const std::vector<lbmvox::Triangle>& tr = mesh.triangles();
        std::vector<VoxTri> vt(tr.size());
        for (size_t i = 0; i < tr.size(); i++) {
            const lbmvox::Triangle& t = tr[i];
            vt[i] = {t.v0.x, t.v0.y, t.v0.z, t.v1.x, t.v1.y, t.v1.z, t.v2.x, t.v2.y, t.v2.z,
                     std::min({t.v0.y, t.v1.y, t.v2.y}), std::max({t.v0.y, t.v1.y, t.v2.y}),
                     std::min({t.v0.z, t.v1.z, t.v2.z}), std::max({t.v0.z, t.v1.z, t.v2.z})};
        }
        VoxFrame fr;
        fr.gcx = g.W / 2.0f; fr.gcy = g.H / 2.0f; fr.gcz = g.Dg / 2.0f;   // mesh_voxelizer.cpp:148-153
        fr.mcx = (lo.x + hi.x) / 2;
        fr.mcy = lo.y + fr.gcy / scale;
        fr.mcz = (lo.z + hi.z) / 2;
        fr.scale = scale;
        VoxTri* d_tris = nullptr;
        int* d_over = nullptr;
        HIPCHK(c, hipMalloc(&d_tris, std::max<size_t>(1, vt.size()) * sizeof(VoxTri)));
        hipError_t e = hipMalloc(&d_over, sizeof(int));
        int over = 0;
        if (e == hipSuccess) e = hipMemsetAsync(d_over, 0, sizeof(int), c->stream);
        if (e == hipSuccess && !vt.empty()) e = hipMemcpyAsync(d_tris, vt.data(), vt.size() * sizeof(VoxTri), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_voxelize_rows, dim3((g.H + 255) / 256, g.D), dim3(256), 0, c->stream, d_tris, (int)vt.size(), fr, c->mask, g, d_over);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&over, d_over, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)hipFree(d_tris); (void)hipFree(d_over);
        if (e != hipSuccess) return fail(c, LBM_ERR_HIP, "lbm_load_mesh: %s", hipGetErrorString(e));
        c->mask_dirty = true;
        on_device = over == 0;  // a ray with more hits than the kernel keeps: redo on the host
    }
    if (!on_device) {
        std::vector<uint8_t> host((size_t)g.W * g.H * g.Dg, 0);
        mesh.voxelize(host.data(), g.W, g.H, g.Dg, scale);
        for (int z = 0; z < g.Dg; z++) memset(host.data() + (size_t)z * g.W * g.H, 255, (size_t)g.W);
        rc = lbm_set_solid_u8(c, host.data() + (size_t)g.z0 * g.W * g.H, (size_t)cells(c));
        if (rc) return rc;
    }
    if (!loaded) return fail(c, LBM_ERR_IO, "lbm_load_mesh: cannot read '%s' (floor-only mask installed, as the reference leaves it)", obj_path);
    return LBM_OK;
}

int lbm_step(lbm_ctx* c, int nsteps) {
    NEED(c);
    if (nsteps < 0) return fail(c, LBM_ERR_INVALID, "lbm_step: nsteps %d", nsteps);
    if (nsteps == 0) return LBM_OK;
    if (c->split_phase != 0) return fail(c, LBM_ERR_STATE, "lbm_step: a split step is in flight (lbm_step_finish missing)");
    int rc = refresh_mask_state(c);
    if (rc) return rc;
    if (c->scheme == LBM_SCHEME_AA) {
        // in place there is no previous buffer to recompute rho/u from: the last step of the batch
        // writes them (20 B/cell once per call)
        rc = ensure_field_buffers(c);
        for (int s = 0; s < nsteps && !rc; s++) rc = launch_aa_step(c, s == nsteps - 1 && c->g.fnz > 0);
        if (rc) return rc;
        c->p_last = c->p;
        c->steps += (uint64_t)nsteps;
        c->fields = FIELDS_VALID;
        c->curl_valid = false;
        return LBM_OK;
    }
    for (int s = 0; s < nsteps; s++) {
        rc = launch_stream_collide<0>(c, c->f[c->cur], c->f[1 - c->cur], c->p, 1, c->g.D);
        if (rc) return rc;
        c->cur = 1 - c->cur;  // std::swap(d_f, d_f_new), lbm_solver.cu:490
        c->zrev ^= 1;
    }
    c->p_last = c->p;
    c->steps += (uint64_t)nsteps;
    c->fields = FIELDS_STALE;
    c->curl_valid = false;
    return LBM_OK;
}

int lbm_compute_curl(lbm_ctx* c) {
    NEED(c);
    int rc = refresh_mask_state(c);
    if (rc) return rc;
    rc = materialise_fields(c);
    if (rc) return rc;
    if (c->curl_valid) return LBM_OK;
    if (c->g.fnz != c->g.D)
        return fail(c, LBM_ERR_STATE, "lbm_compute_curl: the curl's z differences need the whole domain in the field window");
    if (!c->curl) HIPCHK(c, hipMalloc(&c->curl, cells(c) * sizeof(float)));
    dim3 grid((c->g.W + 255) / 256, c->g.H, c->g.D);
    hipLaunchKernelGGL(k_curl, grid, dim3(256), 0, c->stream, c->u, c->curl, c->g.W, c->g.H, c->g.D);
    HIPCHK(c, hipGetLastError());
    c->curl_valid = true;
    return LBM_OK;
}

int lbm_set_field_window(lbm_ctx* c, int z_begin, int nz) {
    NEED(c);
    if (z_begin < 0 || nz < 0 || (long long)z_begin + nz > c->g.D)
        return fail(c, LBM_ERR_INVALID, "lbm_set_field_window: [%d, %d+%d) outside the %d owned layers", z_begin, z_begin, nz, c->g.D);
    if (z_begin == c->g.fz0 && nz == c->g.fnz) return LBM_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->rho); (void)hipFree(c->u); (void)hipFree(c->curl);
    c->rho = nullptr; c->u = nullptr; c->curl = nullptr;
    c->g.fz0 = z_begin; c->g.fnz = nz;
    if (c->fields == FIELDS_VALID) c->fields = FIELDS_STALE;  // A/B: recomputed on demand; AA: after the next step
    c->curl_valid = false;
    return LBM_OK;
}

int lbm_set_streams(lbm_ctx* c, void* stream, void* boundary_stream) {
    NEED(c);
    if (!stream) return fail(c, LBM_ERR_INVALID, "lbm_set_streams: stream is null");
    if (c->split_phase != 0) return fail(c, LBM_ERR_STATE, "lbm_set_streams: a split step is in flight");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->stream_b) HIPCHK(c, hipStreamSynchronize(c->stream_b));
    if (c->own_stream) {
        HIPCHK(c, hipStreamDestroy(c->stream));
        c->own_stream = false;
    }
    c->stream = (hipStream_t)stream;
    c->stream_b = (hipStream_t)boundary_stream;
    return LBM_OK;
}

int lbm_sync(lbm_ctx* c) {
    NEED(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LBM_OK;
}

static int read_back(lbm_ctx* c, void* dst, const void* src, size_t bytes) {
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LBM_OK;
}

int lbm_read_density(lbm_ctx* c, float* rho, size_t n) {
    NEED(c);
    if (!rho || n != (size_t)field_cells(c)) return fail(c, LBM_ERR_INVALID, "lbm_read_density: need %lld floats", field_cells(c));
    int rc = refresh_mask_state(c);
    if (!rc) rc = materialise_fields(c);
    if (rc) return rc;
    return read_back(c, rho, c->rho, n * sizeof(float));
}

int lbm_read_velocity(lbm_ctx* c, float* u, size_t n) {
    NEED(c);
    if (!u || n != (size_t)field_cells(c)) return fail(c, LBM_ERR_INVALID, "lbm_read_velocity: need %lld x 4 floats", field_cells(c));
    int rc = refresh_mask_state(c);
    if (!rc) rc = materialise_fields(c);
    if (rc) return rc;
    return read_back(c, u, c->u, n * sizeof(float4));
}

int lbm_read_curl(lbm_ctx* c, float* curl, size_t n) {
    NEED(c);
    if (!curl || n != (size_t)cells(c)) return fail(c, LBM_ERR_INVALID, "lbm_read_curl: need %lld floats", cells(c));
    int rc = lbm_compute_curl(c);
    if (rc) return rc;
    return read_back(c, curl, c->curl, n * sizeof(float));
}

int lbm_read_velocity_strided(lbm_ctx* c, int sx, int sy, int sz, float* u, size_t n) {
    NEED(c);
    if (sx < 1 || sy < 1 || sz < 1) return fail(c, LBM_ERR_INVALID, "lbm_read_velocity_strided: strides must be positive");
    const int nx = (c->g.W + sx - 1) / sx, ny = (c->g.H + sy - 1) / sy, nz = (c->g.fnz + sz - 1) / sz;
    const size_t need = (size_t)nx * ny * nz;
    if (!u || n != need) return fail(c, LBM_ERR_INVALID, "lbm_read_velocity_strided: need %zu x 4 floats (%d x %d x %d samples)", need, nx, ny, nz);
    int rc = refresh_mask_state(c);
    if (!rc) rc = materialise_fields(c);
    if (rc) return rc;
    float4* d = nullptr;
    HIPCHK(c, hipMalloc(&d, need * sizeof(float4)));
    hipLaunchKernelGGL(k_sample_velocity, grid1((long long)need), dim3(256), 0, c->stream, c->u, d, c->g.W, c->g.H, nx, ny, nz, sx, sy, sz);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(u, d, need * sizeof(float4), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, LBM_ERR_HIP, "lbm_read_velocity_strided: %s", hipGetErrorString(e));
    return LBM_OK;
}

int lbm_read_speed(lbm_ctx* c, float* speed, size_t n) {
    NEED(c);
    if (!speed || n != (size_t)field_cells(c)) return fail(c, LBM_ERR_INVALID, "lbm_read_speed: need %lld floats", field_cells(c));
    int rc = refresh_mask_state(c);
    if (!rc) rc = materialise_fields(c);
    if (rc) return rc;
    float* d = nullptr;
    HIPCHK(c, hipMalloc(&d, n * sizeof(float)));
    hipLaunchKernelGGL(k_speed, grid1((long long)n), dim3(256), 0, c->stream, c->u, d, (long long)n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(speed, d, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, LBM_ERR_HIP, "lbm_read_speed: %s", hipGetErrorString(e));
    return LBM_OK;
}

int lbm_read_solid_u8(lbm_ctx* c, uint8_t* mask, size_t n) {
    NEED(c);
    if (!mask || n != (size_t)cells(c)) return fail(c, LBM_ERR_INVALID, "lbm_read_solid_u8: need %lld bytes", cells(c));
    unsigned char* dense = nullptr;
    HIPCHK(c, hipMalloc(&dense, n));
    hipLaunchKernelGGL(k_unpack_mask, grid1((long long)n), dim3(256), 0, c->stream, c->mask, dense, c->g);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(mask, dense, n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dense);
    if (e != hipSuccess) return fail(c, LBM_ERR_HIP, "lbm_read_solid_u8: %s", hipGetErrorString(e));
    return LBM_OK;
}

int lbm_read_classification(lbm_ctx* c, uint8_t* cls, size_t n19) {
    NEED(c);
    const size_t n = (size_t)cells(c);
    if (!cls || n19 != n * Q) return fail(c, LBM_ERR_INVALID, "lbm_read_classification: need %zu bytes", n * Q);
    unsigned char* d = nullptr;
    HIPCHK(c, hipMalloc(&d, n19));
    if (c->preset == LBM_PRESET_CAVITY)
        hipLaunchKernelGGL(k_classify<LBM_PRESET_CAVITY>, grid1((long long)n), dim3(256), 0, c->stream, c->mask, d, c->g);
    else
        hipLaunchKernelGGL(k_classify<LBM_PRESET_WIND_TUNNEL_REF>, grid1((long long)n), dim3(256), 0, c->stream, c->mask, d, c->g);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(cls, d, n19, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, LBM_ERR_HIP, "lbm_read_classification: %s", hipGetErrorString(e));
    return LBM_OK;
}

// populations <-> reference layout: one strided 2-D copy per plane (rows of W floats out of Wp)
static int copy_populations(lbm_ctx* c, float* host, bool to_host) {
    const Geo& g = c->g;
    const size_t n = (size_t)cells(c);
    for (int k = 0; k < Q; k++) {
        float* dev = c->f[c->cur] + (long long)k * g.plane + g.layer;  // first owned layer
        float* h = host + (size_t)k * n;
        if (to_host)
            HIPCHK(c, hipMemcpy2DAsync(h, (size_t)g.W * 4, dev, (size_t)g.row * 4, (size_t)g.W * 4, (size_t)g.H * g.D, hipMemcpyDeviceToHost, c->stream));
        else
            HIPCHK(c, hipMemcpy2DAsync(dev, (size_t)g.row * 4, h, (size_t)g.W * 4, (size_t)g.W * 4, (size_t)g.H * g.D, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LBM_OK;
}

static int aa_copy_populations(lbm_ctx* c, float* host, bool to_host) {
    const size_t bytes = (size_t)cells(c) * Q * sizeof(float);
    float* dense = nullptr;
    HIPCHK(c, hipMalloc(&dense, bytes));
    hipError_t e = hipSuccess;
    if (to_host) {
        hipLaunchKernelGGL(k_aa_export, grid1(cells(c)), dim3(256), 0, c->stream, c->f[0], dense, c->g, c->aa_phase);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(host, dense, bytes, hipMemcpyDeviceToHost, c->stream);
    } else {
        e = hipMemcpyAsync(dense, host, bytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_aa_import, grid1(cells(c)), dim3(256), 0, c->stream, c->f[0], c->ob[0], c->ob[1], dense, c->g);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dense);
    if (e != hipSuccess) return fail(c, LBM_ERR_HIP, "population copy: %s", hipGetErrorString(e));
    if (!to_host) c->aa_phase = 0;
    return LBM_OK;
}

int lbm_read_populations(lbm_ctx* c, float* f, size_t n19) {
    NEED(c);
    if (!f || n19 != (size_t)cells(c) * Q) return fail(c, LBM_ERR_INVALID, "lbm_read_populations: need %lld floats", cells(c) * Q);
    if (c->scheme == LBM_SCHEME_AA) {
        int rc = refresh_mask_state(c);
        return rc ? rc : aa_copy_populations(c, f, true);
    }
    return copy_populations(c, f, true);
}

int lbm_write_populations(lbm_ctx* c, const float* f, size_t n19) {
    NEED(c);
    if (!f || n19 != (size_t)cells(c) * Q) return fail(c, LBM_ERR_INVALID, "lbm_write_populations: need %lld floats", cells(c) * Q);
    if (c->scheme == LBM_SCHEME_AA) return aa_copy_populations(c, const_cast<float*>(f), false);
    int rc = LBM_OK;
    if (c->fields == FIELDS_STALE) {  // settle rho/u of the last step before touching solid cells
        rc = refresh_mask_state(c);
        if (!rc) rc = materialise_fields(c);
        if (rc) return rc;
    }
    rc = copy_populations(c, const_cast<float*>(f), false);
    if (rc) return rc;
    c->mask_dirty = true;  // solid cells of the other buffer must follow
    return LBM_OK;
}

int lbm_device_fields(lbm_ctx* c, float** d_rho, float** d_u, float** d_curl, uint8_t** d_solid) {
    NEED(c);
    int rc;
    if (c->g.fnz == c->g.D) {
        rc = lbm_compute_curl(c);
    } else {  // partial field window: rho/u of the window only, no curl
        rc = refresh_mask_state(c);
        if (!rc) rc = materialise_fields(c);
    }
    if (rc) return rc;
    if (d_solid) {  // dense copy of the bit-packed mask in the reference's wire format
        const long long n = cells(c);
        if (!c->solid_dense) HIPCHK(c, hipMalloc(&c->solid_dense, (size_t)n));
        hipLaunchKernelGGL(k_unpack_mask, grid1(n), dim3(256), 0, c->stream, c->mask, c->solid_dense, c->g);
        HIPCHK(c, hipGetLastError());
        *d_solid = c->solid_dense;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (d_rho) *d_rho = c->rho;
    if (d_u) *d_u = reinterpret_cast<float*>(c->u);
    if (d_curl) *d_curl = c->curl;
    return LBM_OK;
}

int lbm_get_dims(lbm_ctx* c, int* w, int* h, int* d) {
    if (!c) return fail(nullptr, LBM_ERR_INVALID, "null context");
    if (w) *w = c->g.W;
    if (h) *h = c->g.H;
    if (d) *d = c->g.D;
    return LBM_OK;
}

int lbm_get_stats(lbm_ctx* c, lbm_stats* out) {
    if (!c || !out) return fail(c, LBM_ERR_INVALID, "lbm_get_stats: null argument");
    out->steps = c->steps;
    out->cells = (uint64_t)cells(c);
    out->bytes_per_cell = 2ull * Q * sizeof(float);
    out->population_bytes = (c->scheme == LBM_SCHEME_AA ? 1ull : 2ull) * c->pop_bytes;
    out->last_timed_ms = c->last_ms;
    out->last_timed_steps = c->last_steps;
    return LBM_OK;
}

int lbm_timer_begin(lbm_ctx* c) {
    NEED(c);
    int rc = refresh_mask_state(c);  // keep one-off set-up kernels out of the timed region
    if (rc) return rc;
    c->steps_at_begin = c->steps;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    return LBM_OK;
}

int lbm_timer_end(lbm_ctx* c) {
    NEED(c);
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0.0f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_ms = ms;
    c->last_steps = c->steps - c->steps_at_begin;
    return LBM_OK;
}

// ---- host-only entry points (no device needed) -----------------------------------------------
int lbm_host_voxelize_obj(const char* obj_path, int width, int height, int depth, uint8_t* mask, size_t n, float* scale_out) {
    if (!obj_path || !mask || width < 1 || height < 1 || depth < 1 || n != (size_t)width * height * depth)
        return fail(nullptr, LBM_ERR_INVALID, "lbm_host_voxelize_obj: bad argument");
    std::vector<uint8_t> host;
    const bool loaded = lbmvox::build_wind_tunnel_mask(obj_path, width, height, depth, host, scale_out);
    memcpy(mask, host.data(), n);
    return loaded ? LBM_OK : fail(nullptr, LBM_ERR_IO, "lbm_host_voxelize_obj: cannot read '%s'", obj_path);
}

int lbm_host_collide_cell(const float* f19, float omega_shear, float omega_bulk, float* out19, float* rho_u4) {
    if (!f19 || !out19) return fail(nullptr, LBM_ERR_INVALID, "lbm_host_collide_cell: null argument");
    const Moments m = moments_of(f19);
    collide_cell(f19, m, omega_shear, omega_bulk, out19);
    if (rho_u4) { rho_u4[0] = m.rho; rho_u4[1] = m.ux; rho_u4[2] = m.uy; rho_u4[3] = m.uz; }
    return LBM_OK;
}

// ---- z-slab API ---------------------------------------------------------------------------
static inline size_t halo_tail_floats(const lbm_ctx* c) {
    const size_t n = c->scheme == LBM_SCHEME_AA ? (size_t)OB_N * c->g.H : (size_t)c->g.H + 2;
    return (n + 31) / 32 * 32;
}
// AA: is the buffer in the streamed layout once the step in flight (if any) is complete?
static inline int aa_streamed_now(const lbm_ctx* c) { return (c->split_phase ? c->aa_phase ^ 1 : c->aa_phase) == 0; }
static inline float* aa_newest_ob(lbm_ctx* c) { return c->split_phase ? c->ob[1 - c->ob_cur] : c->ob[c->ob_cur]; }
static int aa_halo(lbm_ctx* c, int side, float* msg, bool pack, hipStream_t stream = nullptr) {
    const long long per = c->g.mlayer > (long long)OB_N * c->g.H ? c->g.mlayer : (long long)OB_N * c->g.H;
    if (!stream) stream = c->stream;
    if (pack)
        hipLaunchKernelGGL(k_aa_halo_pack, dim3(blocks_for(per, 256), 6), dim3(256), 0, stream, c->f[0], aa_newest_ob(c), msg, c->g, side, aa_streamed_now(c));
    else
        hipLaunchKernelGGL(k_aa_halo_unpack, dim3(blocks_for(per, 256), 6), dim3(256), 0, stream, c->f[0], aa_newest_ob(c), msg, c->g, side, aa_streamed_now(c));
    HIPCHK(c, hipGetLastError());
    return LBM_OK;
}

int lbm_halo_message_floats(lbm_ctx* c, size_t* nfloats) {
    if (!c || !nfloats) return fail(c, LBM_ERR_INVALID, "lbm_halo_message_floats: null argument");
    *nfloats = 5 * (size_t)c->g.mlayer + halo_tail_floats(c);
    return LBM_OK;
}

static int halo_args_ok(lbm_ctx* c, int side, const void* p, const char* who) {
    if (!c->g.slab) return fail(c, LBM_ERR_STATE, "%s: context was not created as a slab (lbm_opts.depth_global = 0)", who);
    if (side != 0 && side != 1) return fail(c, LBM_ERR_INVALID, "%s: side must be 0 or 1", who);
    if (!p) return fail(c, LBM_ERR_INVALID, "%s: buffer is null", who);
    return LBM_OK;
}

int lbm_halo_pack(lbm_ctx* c, int side, float* d_message) {
    NEED(c);
    int rc = halo_args_ok(c, side, d_message, "lbm_halo_pack");
    if (rc) return rc;
    if (c->scheme == LBM_SCHEME_AA) return aa_halo(c, side, d_message, true);
    const long long per = c->g.mlayer > c->g.H + 2 ? c->g.mlayer : c->g.H + 2;
    // after lbm_step_interior_layers the new state is still f[1-cur] until lbm_step_finish
    const float* src = c->split_phase ? c->f[1 - c->cur] : c->f[c->cur];
    hipLaunchKernelGGL(k_halo_pack, dim3(blocks_for(per, 256), 6), dim3(256), 0, c->stream, src, d_message, c->g, side);
    HIPCHK(c, hipGetLastError());
    return LBM_OK;
}

int lbm_halo_unpack(lbm_ctx* c, int side, const float* d_message) {
    NEED(c);
    int rc = halo_args_ok(c, side, d_message, "lbm_halo_unpack");
    if (rc) return rc;
    if (c->scheme == LBM_SCHEME_AA) return aa_halo(c, side, const_cast<float*>(d_message), false);
    const long long per = c->g.mlayer > c->g.H + 2 ? c->g.mlayer : c->g.H + 2;
    float* dst = c->split_phase ? c->f[1 - c->cur] : c->f[c->cur];
    hipLaunchKernelGGL(k_halo_unpack, dim3(blocks_for(per, 256), 6), dim3(256), 0, c->stream, dst, d_message, c->g, side);
    HIPCHK(c, hipGetLastError());
    return LBM_OK;
}

int lbm_halo_get_mask_layer(lbm_ctx* c, int side, uint8_t* layer, size_t n) {
    NEED(c);
    int rc = halo_args_ok(c, side, layer, "lbm_halo_get_mask_layer");
    if (rc) return rc;
    const size_t wh = (size_t)c->g.W * c->g.H;
    if (n != wh) return fail(c, LBM_ERR_INVALID, "lbm_halo_get_mask_layer: need %zu bytes", wh);
    unsigned char* d = nullptr;
    HIPCHK(c, hipMalloc(&d, wh));
    hipLaunchKernelGGL(k_mask_layer_get, dim3(blocks_for((long long)wh, 256)), dim3(256), 0, c->stream, c->mask, d, c->g, side ? c->g.D : 1);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(layer, d, wh, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, LBM_ERR_HIP, "lbm_halo_get_mask_layer: %s", hipGetErrorString(e));
    return LBM_OK;
}

int lbm_halo_set_mask_ghost(lbm_ctx* c, int side, const uint8_t* layer, size_t n) {
    NEED(c);
    int rc = halo_args_ok(c, side, layer, "lbm_halo_set_mask_ghost");
    if (rc) return rc;
    const size_t wh = (size_t)c->g.W * c->g.H;
    if (n != wh) return fail(c, LBM_ERR_INVALID, "lbm_halo_set_mask_ghost: need %zu bytes", wh);
    rc = before_mask_change(c);
    if (rc) return rc;
    unsigned char* d = nullptr;
    HIPCHK(c, hipMalloc(&d, wh));
    hipError_t e = hipMemcpyAsync(d, layer, wh, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_mask_layer_set, dim3(blocks_for((long long)c->g.H * c->g.Wm, 256)), dim3(256), 0, c->stream, c->mask, d, c->g, side ? c->g.D + 1 : 0);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, LBM_ERR_HIP, "lbm_halo_set_mask_ghost: %s", hipGetErrorString(e));
    c->mask_dirty = true;
    return LBM_OK;
}

int lbm_step_boundary_layers(lbm_ctx* c, float* d_msg_bottom, float* d_msg_top) {
    NEED(c);
    if (c->split_phase != 0) return fail(c, LBM_ERR_STATE, "lbm_step_boundary_layers: previous split step not finished");
    if ((d_msg_bottom || d_msg_top) && !c->g.slab)
        return fail(c, LBM_ERR_STATE, "lbm_step_boundary_layers: messages need a slab context");
    const bool was_dirty = c->mask_dirty;
    int rc = refresh_mask_state(c);
    if (rc) return rc;
    if (was_dirty && c->stream_b) {  // the set-up kernels ran on the main stream: order the side stream behind them
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream_b, c->ev1, 0));
    }
    if (c->scheme == LBM_SCHEME_AA) {
        // in place: the boundary layers, then two small pack launches (the layout of the moment decides
        // which slots travel, see k_aa_halo_pack); rho/u are written when the field window is not empty
        hipStream_t bs = c->stream_b ? c->stream_b : c->stream;  // boundary layers + packs feed the exchange
        rc = ensure_field_buffers(c);
        if (!rc) rc = launch_aa(c, c->g.fnz > 0, 1, c->g.D > 1 ? 2 : 1, c->g.D, bs);
        if (rc) return rc;
        c->split_phase = 1;
        if (d_msg_bottom) rc = aa_halo(c, 0, d_msg_bottom, true, bs);
        if (!rc && d_msg_top) rc = aa_halo(c, 1, d_msg_top, true, bs);
        return rc;
    }
    // one launch for both boundary layers; it writes the outgoing messages as it goes
    rc = launch_stream_collide<2>(c, c->f[c->cur], c->f[1 - c->cur], c->p, 1, c->g.D > 1 ? 2 : 1, c->g.D,
                                  d_msg_bottom, d_msg_top);
    if (rc) return rc;
    c->split_phase = 1;
    return LBM_OK;
}

int lbm_halo_unpack_both(lbm_ctx* c, const float* d_from_below, const float* d_from_above) {
    NEED(c);
    if (!d_from_below && !d_from_above) return LBM_OK;
    int rc = halo_args_ok(c, 0, d_from_below ? d_from_below : d_from_above, "lbm_halo_unpack_both");
    if (rc) return rc;
    if (c->scheme == LBM_SCHEME_AA) {
        if (d_from_below) rc = aa_halo(c, 0, const_cast<float*>(d_from_below), false);
        if (!rc && d_from_above) rc = aa_halo(c, 1, const_cast<float*>(d_from_above), false);
        return rc;
    }
    const long long per = c->g.mlayer > c->g.H + 2 ? c->g.mlayer : c->g.H + 2;
    float* dst = c->split_phase ? c->f[1 - c->cur] : c->f[c->cur];
    hipLaunchKernelGGL(k_halo_unpack2, dim3(blocks_for(per, 256), 12), dim3(256), 0, c->stream, dst, d_from_below, d_from_above, c->g);
    HIPCHK(c, hipGetLastError());
    return LBM_OK;
}

int lbm_step_interior_layers(lbm_ctx* c) {
    NEED(c);
    if (c->split_phase != 1) return fail(c, LBM_ERR_STATE, "lbm_step_interior_layers: call lbm_step_boundary_layers first");
    int rc = c->scheme == LBM_SCHEME_AA ? launch_aa(c, c->g.fnz > 0, 2, c->g.D - 2, 0)
                                        : launch_stream_collide<0>(c, c->f[c->cur], c->f[1 - c->cur], c->p, 2, c->g.D - 2);
    if (rc) return rc;
    c->split_phase = 2;
    return LBM_OK;
}

int lbm_step_finish(lbm_ctx* c) {
    NEED(c);
    if (c->split_phase != 2) return fail(c, LBM_ERR_STATE, "lbm_step_finish: call lbm_step_interior_layers first");
    if (c->scheme == LBM_SCHEME_AA) aa_flip(c);
    else c->cur = 1 - c->cur;
    c->p_last = c->p;
    c->steps += 1;
    c->fields = c->scheme == LBM_SCHEME_AA ? FIELDS_VALID : FIELDS_STALE;
    c->curl_valid = false;
    c->split_phase = 0;
    return LBM_OK;
}

}  // extern "C"
