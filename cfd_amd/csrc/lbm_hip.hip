// lbm_hip.hip -- MI355X (gfx950) D3Q19 lattice-Boltzmann hot path behind include/lbm_abi.h.
//
// Replaces Engine/src/Physics/lbm_solver.cu of the reference (kernels :37-64, :242-402,
// :498-513, :612-656 and the host methods around them).  Written for CDNA4 from scratch:
//   * populations: SoA fp32 runs of Wp floats (W rounded up to 32 floats: every run starts on a
//     128-byte line), D+2 z-layers (one ghost layer each side for the slab halo).  Default layout is
//     row-interleaved [z][y][k][x] -- the 19 runs of one (y,z) row form one contiguous chunk --, the
//     19 skewed planes [k][z][y][x] of round 1a stay selectable; every kernel addresses through the
//     strides of struct Geo (k_step.h), so the layout changes no result;
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
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lbm_abi.h"
#include "d3q19.h"
#include "voxelizer.h"

using namespace d3q19;

namespace {

#include "k_step.h"
#include "k_fused2.h"
#include "k_aa.h"
#include "k_setup.h"
#include "k_voxelize.h"
#include "k_mask.h"
#include "k_halo.h"
#include "k_fields.h"

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
    int slide_chunk = 0;     // SLIDE: layers per launch; the lattice moves by slide_chunk + 2 layers per step
    bool f1_in_alloc0 = false;
    bool xcd_rows = false;     // ZORDER_XCD: whole rows of workgroups per XCD (k_step.h tile_of_wave)
    bool probe = false;        // LBM_PROBE=1: launch knobs are re-read from the environment at every lbm_step (tuning probes)
    int aa_chunk = 0;        // AA: layers per launch (0 = the whole domain in one launch)
    int ab_chunk = 0;        // A/B: layers per launch (0 = the whole domain in one launch)
    long long slide_off = 0;   // SLIDE: offset (floats) of the lattice inside the buffer right now: 0 or slide_shift
    long long slide_shift = 0; // SLIDE: (chunk + 2) layers + slide_skew
    bool fields_wanted = false;   // A/B: the caller read rho/u since its last lbm_step() call (fuse them into the next one)
    bool slide_pristine = false;  // SLIDE: every solid cell holds w_k (init() wrote it, no fluid cell turned solid after a
                                  // step): solid cells are skipped instead of carried along (k_stream_collide MODE 3)
    bool slide_stepped = false;   // SLIDE: a step ran since init(): skipped solid cells' slots hold stale values
    long long slide_skew = 0;  // SLIDE: extra floats between the two positions (DRAM bank alignment of loads against stores)
    float* alloc0 = nullptr;   // AA: start of the allocation (f[0] sits behind a front guard)
    float* ob[2] = {nullptr, nullptr};
    int ob_cur = 0;
    unsigned long long* prev_mask = nullptr;
    size_t mask_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // lbm_timer_begin / lbm_timer_end
    hipEvent_t ev_order = nullptr;            // orders the boundary stream behind / ahead of the main stream (split step)
    bool slide_wait_b = false;                // SLIDE split step: the interior's second chunk must wait for the boundary launches
    bool prev_touched = false;                // A/B: the previous buffer or the mask changed since the last step
    uint64_t steps = 0, steps_at_begin = 0;
    uint64_t launches = 0, launches_at_begin = 0, last_launches = 0;  // step-kernel launches (a step may take several)
    double last_ms = 0.0;
    uint64_t last_steps = 0;
    size_t pop_bytes = 0;
    long long pop_floats = 0;  // floats of one population buffer behind f[i]
    long long guard_floats = 0;  // AA: floats allocated in front of f[0]
    int layout = LBM_LAYOUT_ROWS;
    unsigned lds_pad = 0;      // dynamic LDS bytes per block of the step kernel: caps the blocks per CU (occupancy throttle)
    bool pair_wave_blocks = true;  // the boundary layers' second step of a pair in one-wave workgroups (ZORDER_WAVE; LBM_PAIR_WAVE_BLOCKS=0: off)
    bool wave_blocks_now = false;  // ... for the launch being issued
    // ---- two steps per pass (k_fused2.h; A/B, undecomposed, row-interleaved layout) ----
    bool fused_on = false;         // lbm_opts.two_step / LBM_FUSED
    bool fused_ready = false;      // eligibility map and rim lists match the current mask
    bool fused_usable = false;     // ... and enough of the domain takes the fused path for it to pay
    int f2_ty = 12, f2_stage = 1, f2_nbuf = 1;  // kernel variant: measured best of the compiled ones (profiles/r02_tuning.md) (LBM_FUSED_VARIANT="TY,STAGE,NBUF")
    int f2_zsplit = 0;             // LBM_FUSED_ZSPLIT: z chunks per tile column (0 = chosen by fused_prepare)
    bool f2_obst = true;           // LBM_FUSED_OBST=0: blocks keep clear of obstacles (round-2 first form)
    int f2_ny = 0;                 // LBM_FUSED_NY: y tiles per tile column (0 = chosen by fused_prepare)
    unsigned long long* f2_stamps = nullptr;   // lbm_fused_chain_stamps(): start / end clock of every workgroup of one launch
    bool f2_obst_blocks = false;   // some block of the current list holds or touches solid cells
    bool f2_walls = false;         // LBM_FUSED_WALLS=1: blocks may lie on the domain's walls (WALL form of k_fused2; measured: pays for large cavities only, profiles/r03_tuning.md)
    int f2_wide = -1;              // LBM_FUSED_WIDE: 1 / 0 = plain blocks take / do not take the wide form of the first-step loads; -1 = timed on the first launch after a plan
    bool f2_wide_pending = false;  // the current plan is still being timed (f2_wide == -1)
    std::vector<hipEvent_t> f2_tune_ev;   // start / end events of the tuning launches so far
    int f2_tune_n = 0;
    float f2_wide_ms[2] = {0.f, 0.f};   // what the two forms took when they were timed (narrow, wide)
    bool f2_wall_blocks = false;   // the current list was planned with blocks on the walls: the WALL form of the kernel
    unsigned* f2_nbw = nullptr;    // one word per cell: which directions bounce back from a solid neighbour (k_f2_nb_words)
    FusedGrid fg{};
    float* ft = nullptr;           // third population buffer: holds time t+1 on the rim between its two steps
    unsigned long long* f2_blocks = nullptr;   // the blocks (x tile | y tile << 12 | first layer << 24 | layers << 44), one workgroup of k_fused2 each
    size_t f2_blocks_cap = 0;
    unsigned long long* rim_list[2] = {nullptr, nullptr};   // [0]: tiles that need t+1 (rim + its neighbours), [1]: the rim
    size_t rim_n[2] = {0, 0}, rim_cap[2] = {0, 0};          // entries, padded to a multiple of 4
    long long fused_cells = 0;     // cells one k_fused2 launch updates (twice each)
    long long rim_cells = 0;       // cells of the rim tiles (second list)
    long long tail_cells = 0;      // cells of the rim's first step (first list: the rim tiles and the tiles next to them)
    hipStream_t stream_rim = nullptr;   // library-owned side stream: the rim's two small launches run beside k_fused2
    hipEvent_t ev_f2_main = nullptr, ev_f2_rim = nullptr;
    bool f2_tail_rim = true;       // the rim's first step behind the block list of the fused launch (k_fused2.h F2Rim; LBM_FUSED_TAIL_RIM=0: a launch of its own)
    int f2_tail_pct = 4;           // ... when the rim is more than this share of the cells (LBM_FUSED_TAIL_PCT)
    int f2_rounds = 1;             // rounds of blocks a fused launch runs in (blocks / CUs, rounded up)
    int f2_tail_now = 0;           // ... for the launch being issued: 1 the rim's first step, 2 its second (slabs)
    bool f2_overlap = true;        // LBM_FUSED_OVERLAP=0: everything on the context's stream
    int f2_overlap_pct = 15;       // LBM_FUSED_OVERLAP_PCT: side stream when the rim is more than this share of the cells
    bool f2_rim_pending = false;   // the side stream holds rim launches the context's stream has not waited for yet
    // a PAIR of steps on a z-slab (lbm_pair_*): 0 idle, 1 boundary layers of the first step issued, 2 interior issued,
    // 3 boundary layers of the second step issued
    int pair_phase = 0;
    bool pair_rim_stream = false;  // LBM_PAIR_RIM_STREAM=1: the rim's launches of a pair go to a third stream (measured: no gain)
    bool side_prio = true;         // LBM_SIDE_PRIO=0: the short kernels beside k_fused2 do not raise their waves' priority
    hipEvent_t ev_pair_b1 = nullptr, ev_pair_r1 = nullptr;
    bool pair_last = false;        // the last update was a pair: the previous buffer is TWO steps old (no rho/u on demand)
    bool ended_on_pair = false;    // lbm_step() ended on a fused pair: f[1-cur] is two steps old, rho/u on demand take one step through ft
    std::vector<hipEvent_t> f2_ev; // event pairs around the k_fused2 launches of a timed region
    size_t f2_ev_used = 0;
    bool timing = false;
    uint64_t fused_launches = 0, fused_launches_at_begin = 0, last_fused_launches = 0;
    double last_fused_ms = 0.0;
    std::string err;
};

static thread_local std::string g_create_error;

// Scratch of lbm_read_velocity_strided_noalloc (include/lbm_abi.h): the staging area its sampling kernel
// writes before the copy to the host, held in the code object instead of a hipMalloc per call.  Every
// device the library runs on has its own copy at its own address; `scratch` itself is only the host-side
// handle of the symbol, so host code uses the per-device address that lbm_create() looks up below.
static __device__ __align__(256) unsigned char scratch[LBM_NOALLOC_MAX_SAMPLES * 4 * sizeof(float)];
constexpr int MAX_DEVICES = 64;
static void* g_scratch_ptr[MAX_DEVICES] = {nullptr};
// once per device, with that device current
static hipError_t resolve_scratch(int dev) {
    static std::mutex lock;                        // contexts may be created from several host threads at once
    std::lock_guard<std::mutex> hold(lock);
    if (dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    if (g_scratch_ptr[dev]) return hipSuccess;
    void* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e == hipSuccess) g_scratch_ptr[dev] = ptr;
    return e;
}

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

// the populations of the current time level (the sliding scheme keeps them at a layer offset)
static inline float* current_populations(lbm_ctx* c) {
    return c->scheme == LBM_SCHEME_SLIDE ? c->f[0] + c->slide_off : c->f[c->cur];
}

static int ensure_field_buffers(lbm_ctx* c) {
    const long long n = field_cells(c);
    if (n == 0) return LBM_OK;
    if (!c->rho) HIPCHK(c, hipMalloc(&c->rho, n * sizeof(float)));
    if (!c->u) HIPCHK(c, hipMalloc(&c->u, n * sizeof(float4)));
    return LBM_OK;
}

static int slide_fill_solids(lbm_ctx* c, const unsigned long long* mask);

// rebuild tile flags and re-freeze solid cells after a mask / population change
static int refresh_mask_state(lbm_ctx* c) {
    if (!c->mask_dirty) return LBM_OK;
    const long long ntiles = (long long)c->g.D * c->g.H * c->g.Wt;
    if (c->preset == LBM_PRESET_CAVITY)
        hipLaunchKernelGGL(k_tile_flags<LBM_PRESET_CAVITY>, grid1(ntiles), dim3(256), 0, c->stream, c->mask, c->flags, c->g);
    else
        hipLaunchKernelGGL(k_tile_flags<LBM_PRESET_WIND_TUNNEL_REF>, grid1(ntiles), dim3(256), 0, c->stream, c->mask, c->flags, c->g);
    HIPCHK(c, hipGetLastError());
    if (getenv("LBM_FORCE_GENERAL"))  // probe: every tile takes the boundary path (same results, measures its cost)
        HIPCHK(c, hipMemsetAsync(c->flags, 1, (size_t)ntiles, c->stream));
    if (c->scheme == LBM_SCHEME_AA) {
        hipLaunchKernelGGL(k_aa_refluidize, grid1(cells(c)), dim3(256), 0, c->stream, c->f[0],
                           c->ob[c->ob_cur], c->prev_mask, c->mask, c->g, c->aa_phase);
        HIPCHK(c, hipGetLastError());
        hipLaunchKernelGGL(k_ob_sync, grid1((long long)c->g.D * c->g.H), dim3(256), 0, c->stream,
                           c->ob[c->ob_cur], c->ob[1 - c->ob_cur], c->mask, c->g);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(c->prev_mask, c->mask, c->mask_bytes, hipMemcpyDeviceToDevice, c->stream));
    } else if (c->scheme == LBM_SCHEME_SLIDE) {
        // A mask change after a step ends the pristine mode (cells that turn solid now freeze what they
        // hold, no longer w_k): give the cells that were solid so far their w_k back first.  From here on
        // solid cells are carried along by the step itself (k_stream_collide MODE 3).  The outlet
        // column's side array needs the frozen values in both of its buffers.
        if (c->slide_pristine && c->slide_stepped) {
            int rc = slide_fill_solids(c, c->prev_mask);
            if (rc) return rc;
            c->slide_pristine = false;
        }
        HIPCHK(c, hipMemcpyAsync(c->prev_mask, c->mask, c->mask_bytes, hipMemcpyDeviceToDevice, c->stream));
        if (c->ob[0]) {
            hipLaunchKernelGGL(k_ob_sync, grid1((long long)c->g.D * c->g.H), dim3(256), 0, c->stream,
                               c->ob[c->ob_cur], c->ob[1 - c->ob_cur], c->mask, c->g);
            HIPCHK(c, hipGetLastError());
        }
    } else {
        hipLaunchKernelGGL(k_sync_solid, grid1(cells(c)), dim3(256), 0, c->stream,
                           c->f[c->cur], c->f[1 - c->cur], c->mask, c->g);
        HIPCHK(c, hipGetLastError());
        c->prev_touched = true;  // the previous buffer no longer is what the last step read
        if (c->ft) {  // the rim's intermediate buffer keeps the frozen solid cells too (the outlet branch reads them)
            hipLaunchKernelGGL(k_sync_solid, grid1(cells(c)), dim3(256), 0, c->stream, c->f[c->cur], c->ft, c->mask, c->g);
            HIPCHK(c, hipGetLastError());
        }
    }
    c->mask_dirty = false;
    c->fused_ready = false;  // tile flags changed: the fused path's eligibility map and rim lists are stale
    return LBM_OK;
}

template <int MODE>
static int launch_stream_collide(lbm_ctx* c, const float* fin, float* fout, const Params& p, int zl_begin, int nlayers,
                                 int zl_second = 0, float* msg_bottom = nullptr, float* msg_top = nullptr,
                                 int want_moments = 0, hipStream_t on_stream = nullptr,
                                 const unsigned long long* tile_list = nullptr, size_t ntiles = 0) {
    if (nlayers <= 0) return LBM_OK;
    hipStream_t stream = on_stream ? on_stream : ((MODE == 2 && c->stream_b) ? c->stream_b : c->stream);
    const int segsx = 1 << c->segsx_log2;
    dim3 grid((c->g.Wt + segsx - 1) / segsx, (c->g.H + (4 / segsx) - 1) / (4 / segsx), nlayers);
    if (MODE == 4) {  // one wave per list entry, four entries per workgroup
        if (ntiles == 0) return LBM_OK;
        grid = dim3((unsigned)(ntiles / 4));
    }
    if (c->probe) { const char* t = getenv("LBM_XCD_ROWS"); c->xcd_rows = t ? atoi(t) != 0 : false; }
    const bool wave_blocks = MODE == 2 && c->wave_blocks_now;
    if (wave_blocks) grid = dim3(c->g.Wt, c->g.H, nlayers);
    const int zrev = ((MODE == 0 && (c->tune & 4)) ? c->zrev : 0) | ((c->xcd_rows && !wave_blocks) ? ZORDER_XCD : 0) |
                     (((MODE == 2 || MODE == 4) && c->side_prio && c->fused_usable) ? ZORDER_PRIO : 0) | (wave_blocks ? ZORDER_WAVE : 0);
#define LBM_LAUNCH(PRESET, TUNE)                                                                              \
    hipLaunchKernelGGL((k_stream_collide<PRESET, MODE, TUNE>), grid, dim3(wave_blocks ? 64 : 256), c->lds_pad, stream, fin, fout, c->mask, \
                       c->flags, c->rho, c->u, c->g, p, zl_begin, c->segsx_log2, zrev, zl_second, msg_bottom, msg_top,     \
                       c->ob[c->ob_cur], c->ob[1 - c->ob_cur], want_moments, tile_list)
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
    if (MODE != 1 && MODE != 4) c->launches++;
    return LBM_OK;
}

// A/B update of layers [zl_begin, zl_begin + nlayers) in launches of ab_chunk layers (see auto_chunk())
static int launch_ab_layers(lbm_ctx* c, int zl_begin, int nlayers, int want_moments = 0) {
    if (c->probe) { const char* t = getenv("LBM_AB_CHUNK"); c->ab_chunk = t ? std::max(0, atoi(t)) : 0; }
    const int K = c->ab_chunk > 0 ? c->ab_chunk : nlayers;
    const int nchunks = (nlayers + K - 1) / K;
    const bool top_down = (c->tune & 4) && c->zrev;  // the launches follow the layer order inside a launch
    for (int i = 0; i < nchunks; i++) {
        const int zb = (top_down ? nchunks - 1 - i : i) * K;
        int rc = launch_stream_collide<0>(c, c->f[c->cur], c->f[1 - c->cur], c->p, zl_begin + zb, std::min(K, nlayers - zb),
                                          0, nullptr, nullptr, want_moments);
        if (rc) return rc;
    }
    return LBM_OK;
}

// one AA launch over layers [zl_begin, zl_begin + nlayers), or over the two layers zl_begin and
// zl_second when zl_second > 0; the caller flips the phase with aa_flip() once the step is complete
static int launch_aa(lbm_ctx* c, int want_moments, int zl_begin, int nlayers, int zl_second, hipStream_t stream = nullptr,
                     bool whole_step = false) {
    if (nlayers <= 0) return LBM_OK;
    if (!stream) stream = c->stream;
    const int segsx = 1 << c->segsx_log2;
    dim3 grid((c->g.Wt + segsx - 1) / segsx, (c->g.H + (4 / segsx) - 1) / (4 / segsx), nlayers);
    if (c->probe) { const char* t = getenv("LBM_XCD_ROWS"); c->xcd_rows = t ? atoi(t) != 0 : false; }
    const int zrev = (((c->tune & 4) && (whole_step || nlayers == c->g.D)) ? c->zrev : 0) | (c->xcd_rows ? ZORDER_XCD : 0);
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
    c->launches++;
    return LBM_OK;
}
static void aa_flip(lbm_ctx* c) {
    c->aa_phase ^= 1;
    c->ob_cur ^= 1;
    c->zrev ^= 1;
}
// AA update of layers [zl_begin, zl_begin + nlayers) in launches of aa_chunk layers (auto_chunk():
// 1024x512x512 runs 8 % faster in launches of 8-16 layers than in one, 4.95 -> 5.38 TB/s)
static int launch_aa_layers(lbm_ctx* c, int want_moments, int zl_begin, int nlayers) {
    if (c->probe) { const char* t = getenv("LBM_AB_CHUNK"); c->aa_chunk = t ? std::max(0, atoi(t)) : 0; }
    const int K = c->aa_chunk > 0 ? c->aa_chunk : nlayers;
    const int nchunks = (nlayers + K - 1) / K;
    const bool top_down = (c->tune & 4) && c->zrev;
    for (int i = 0; i < nchunks; i++) {
        const int zb = (top_down ? nchunks - 1 - i : i) * K;
        int rc = launch_aa(c, want_moments, zl_begin + zb, std::min(K, nlayers - zb), 0, nullptr, true);
        if (rc) return rc;
    }
    return LBM_OK;
}
static int launch_aa_step(lbm_ctx* c, int want_moments) {
    int rc = launch_aa_layers(c, want_moments, 1, c->g.D);
    if (!rc) aa_flip(c);
    return rc;
}

// Sliding in-place scheme: ONE buffer holds the lattice at a layer offset that alternates between 0
// and S = chunk + 2.  A step reads the lattice at one offset and writes it at the other, chunk by
// chunk in the direction of the shift (lattice at 0 -> written at S: top chunk first; at S ->
// written at 0: bottom chunk first).  A chunk of K <= S-2 layers writes absolute layers that lie
// outside everything it reads ([zb+S, zb+K-1+S] against [zb-1, zb+K]), and the input layers it
// overwrites are only needed by chunks that have already run; launches on one stream serialise.
// So the update has the access pattern of the A/B kernel (aligned stores, pulled loads) with
// 1 + S/D of one buffer instead of two.  The outlet branch, which would read layers the sweep has
// overwritten (quirk Q2 wraps to the far end), goes through the `ob` side array as in AA.
//
// As a z-slab the sweep keeps its direction, so only the boundary layer the sweep STARTS with can be
// updated ahead of the interior; the other one gets a message-only pass first (it reads three input
// layers and writes nothing but its halo message) and its real update after the interior.  The
// incoming messages go into the ghost layers of the NEW lattice position once the interior has
// consumed the input layers that lie there (lbm_halo_unpack_both runs behind the interior).
static inline bool slide_up(const lbm_ctx* c) { return c->slide_off == 0; }
static inline const float* slide_in(const lbm_ctx* c) { return c->f[0] + c->slide_off; }
static inline float* slide_out(const lbm_ctx* c) { return c->f[0] + (slide_up(c) ? c->slide_shift : 0); }
// layers [zl_begin, zl_begin + nlayers) in sweep order, slide_chunk layers per launch
static int slide_layers(lbm_ctx* c, int zl_begin, int nlayers, int want_moments, float* msg_bottom = nullptr,
                        float* msg_top = nullptr) {
    if (nlayers <= 0) return LBM_OK;
    const int K = c->slide_chunk;
    const bool up = slide_up(c);
    const int nchunks = (nlayers + K - 1) / K;
    for (int i = 0; i < nchunks; i++) {
        const int j = up ? nchunks - 1 - i : i;
        const int nl = nlayers - j * K < K ? nlayers - j * K : K;
        // Split step on two streams: the boundary launches (lbm_step_boundary_layers, on the side stream)
        // read the input layers around the layer the sweep starts with, and from the second chunk on the
        // interior's output lands on exactly those layers.  The first chunk still overlaps with them.
        if (i == 1 && c->slide_wait_b) {
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_order, 0));
            c->slide_wait_b = false;
        }
        int rc = launch_stream_collide<3>(c, slide_in(c), slide_out(c), c->p, zl_begin + j * K, nl, 0, msg_bottom, msg_top,
                                          want_moments | (c->slide_pristine ? 2 : 0));
        if (rc) return rc;
    }
    return LBM_OK;
}
static void slide_flip(lbm_ctx* c) {
    c->slide_off = slide_up(c) ? c->slide_shift : 0;
    c->ob_cur ^= 1;
    c->slide_stepped = true;
}
// pristine mode: make the slots of the cells `mask` marks solid valid (w_k) at the current lattice position
static int slide_fill_solids(lbm_ctx* c, const unsigned long long* mask) {
    if (!(c->slide_pristine && c->slide_stepped)) return LBM_OK;
    hipLaunchKernelGGL(k_slide_fill_solids, grid1(cells(c)), dim3(256), 0, c->stream, c->f[0] + c->slide_off, mask, c->g);
    HIPCHK(c, hipGetLastError());
    return LBM_OK;
}
static int slide_step(lbm_ctx* c, int want_moments) {
    int rc = slide_layers(c, 1, c->g.D, want_moments);
    if (!rc) slide_flip(c);
    return rc;
}
// where the halo calls find the lattice: between lbm_step_interior_layers and lbm_step_finish the new
// state already sits at the other position
static inline float* halo_lattice(lbm_ctx* c) {
    if (c->scheme == LBM_SCHEME_SLIDE) return c->split_phase ? slide_out(c) : c->f[0] + c->slide_off;
    return c->split_phase ? c->f[1 - c->cur] : c->f[c->cur];
}

// ------------------------------------------------------------------------------------------
// two steps per pass (k_fused2.h)
// ------------------------------------------------------------------------------------------
// The fused kernel covers the blocks (64 x TY tile, chunk of z layers) whose haloed neighbourhood is plain
// fluid; everything else -- the rim: wall rows and layers, tiles near solids, ragged remainders -- takes two
// ordinary steps through the third buffer `ft`:
//     k_fused2            f[cur] -> f[1-cur]   (time t -> t+2, covered tiles)
//     k_stream_collide<4> f[cur] -> ft         (t -> t+1 on the rim and on every tile next to it)
//     k_stream_collide<4> ft     -> f[1-cur]   (t+1 -> t+2 on the rim)
// The three launches write disjoint cells; the second pair is ordered by the stream.  fused_prepare() derives
// block eligibility and the two tile lists from the tile flags (on the host: it runs once per mask change).
static bool fused_applies(const lbm_ctx* c) {   // lbm_step() on an undecomposed context
    return c->fused_on && c->scheme == LBM_SCHEME_AB && !c->g.slab && c->layout == LBM_LAYOUT_ROWS;
}
static bool fused_applies_slab(const lbm_ctx* c) {   // lbm_pair_*() on a z-slab
    return c->fused_on && c->scheme == LBM_SCHEME_AB && c->g.slab && c->layout == LBM_LAYOUT_ROWS && c->g.D >= 6;
}

struct F2Variant { int ty, stage, nbuf; };
static const F2Variant kF2Variants[] = {{9, 1, 2}, {12, 1, 1}, {13, 1, 1}, {9, 0, 2}, {12, 0, 2}};

template <int PRESET, int TY, int STAGE, int NBUF, int OBST, int WALL>
static hipError_t launch_fused_kernel(lbm_ctx* c, const float* fin, float* fout) {
    static bool attr_set[MAX_DEVICES] = {false};
    auto kern = k_fused2<PRESET, TY, STAGE, NBUF, OBST, WALL>;
    const size_t lds = f2_lds_bytes(TY, STAGE, NBUF);
    if (!attr_set[c->device]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set[c->device] = true;
    }
    // a step of the rim rides behind the block list (F2Rim) when the caller asked for it: 1 = its first step (fin -> ft,
    // undecomposed contexts), 2 = its second (ft -> fout, z-slabs)
    F2Rim rim{nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
    unsigned rim_wgs = 0;
    const int which = c->f2_tail_now - 1;
    if (which >= 0 && c->rim_n[which] > 0) {
        rim = F2Rim{c->rim_list[which], c->mask, c->flags, which == 0 ? fin : c->ft, which == 0 ? c->ft : fout, (unsigned)c->rim_n[which]};
        rim_wgs = (unsigned)((c->rim_n[which] + (TY + 3) - 1) / (TY + 3));
    }
    hipLaunchKernelGGL(kern, dim3(c->fg.nblocks + rim_wgs), dim3((TY + 3) * 64), lds, c->stream, fin, fout, c->f2_blocks,
                       c->f2_nbw, c->g, c->p, c->fg, c->f2_stamps, rim);
    return hipGetLastError();
}
template <int PRESET, int TY, int STAGE, int NBUF>
static hipError_t launch_fused_variant(lbm_ctx* c, const float* fin, float* fout) {
    // blocks that hold solid cells need the bounce-back form of the kernel, blocks on the domain's walls the wall form
    // of that; a domain without either keeps the plain one
    if (c->f2_wall_blocks) return launch_fused_kernel<PRESET, TY, STAGE, NBUF, 1, 1>(c, fin, fout);
    return c->f2_obst_blocks ? launch_fused_kernel<PRESET, TY, STAGE, NBUF, 1, 0>(c, fin, fout)
                             : launch_fused_kernel<PRESET, TY, STAGE, NBUF, 0, 0>(c, fin, fout);
}
template <int TY, int STAGE, int NBUF>
static hipError_t launch_fused_preset(lbm_ctx* c, const float* fin, float* fout) {
    return c->preset == LBM_PRESET_CAVITY ? launch_fused_variant<LBM_PRESET_CAVITY, TY, STAGE, NBUF>(c, fin, fout)
                                          : launch_fused_variant<LBM_PRESET_WIND_TUNNEL_REF, TY, STAGE, NBUF>(c, fin, fout);
}
static hipError_t launch_fused(lbm_ctx* c, const float* fin, float* fout) {
    const int key = c->f2_ty * 100 + c->f2_stage * 10 + c->f2_nbuf;
    switch (key) {
        case 912: return launch_fused_preset<9, 1, 2>(c, fin, fout);
        case 1211: return launch_fused_preset<12, 1, 1>(c, fin, fout);
        case 1311: return launch_fused_preset<13, 1, 1>(c, fin, fout);
        case 902: return launch_fused_preset<9, 0, 2>(c, fin, fout);
        case 1202: return launch_fused_preset<12, 0, 2>(c, fin, fout);
    }
    return hipErrorInvalidValue;
}

static int fused_upload(lbm_ctx* c, void** dev, size_t* cap, const void* host, size_t bytes) {
    if (bytes > *cap) {
        if (*dev) (void)hipFree(*dev);
        *dev = nullptr; *cap = 0;
        HIPCHK(c, hipMalloc(dev, bytes));
        *cap = bytes;
    }
    if (bytes) HIPCHK(c, hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    return LBM_OK;
}

// the mask is clean (refresh_mask_state ran).  Leaves fused_usable = false when the fused path does not apply
// to this domain or would not pay; that is not an error.
static void fused_tune_reset(lbm_ctx* c);
static int fused_prepare(lbm_ctx* c) {
    if (c->fused_ready) return LBM_OK;
    c->fused_ready = true;
    c->fused_usable = false;
    const Geo& g = c->g;
    const int TY = c->f2_ty;
    // rows [2, H-2) and layers z in [2, D-2) can take the fused path: the haloed blocks must stay off the wall
    // rows / layers, whose tiles are flagged as boundary tiles
    if (g.H < 6 || g.D < 6 || g.Wt > 4096 || g.H >= (1 << 24) || g.D + 2 >= (1 << 24)) return LBM_OK;
    const long long layer_b = g.layer * 4;
    if (layer_b >= (1ll << 30)) return LBM_OK;                      // 32-bit byte offsets inside a z chunk (k_fused2.h)
    const int max_chunk = (int)std::min<long long>(g.D, ((1ll << 32) - 1) / layer_b - 5);
    if (max_chunk < 1) return LBM_OK;
    int ncu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount; }
    if (const char* t = getenv("LBM_FUSED_NCU")) ncu = std::max(8, atoi(t));   // tests / probes: plan for another CU count

    // tile flags -> host
    const long long ntiles = (long long)g.D * g.H * g.Wt;
    std::vector<unsigned char> fl((size_t)ntiles);
    HIPCHK(c, hipMemcpyAsync(fl.data(), c->flags, (size_t)ntiles, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    auto tile = [&](int xs, int y, int z) -> size_t { return ((size_t)z * g.H + y) * g.Wt + xs; };   // z = zl - 1

    // The rows the tiles start from: the first row whose lower neighbour is (mostly) plain fluid.  A closed box
    // gives y_first = 2 (row 0 is the wall row, row 1 the lower halo row); the wind tunnel's solid floor plane
    // (lbm_solver.cu:549-553) flags row 1 as well and moves the tiles up by one.
    auto row_clean = [&](int y) {
        long long clean = 0;
        for (int z = 1; z + 1 < g.D; z++)
            for (int xs = 0; xs < g.Wt; xs++) clean += fl[tile(xs, y, z)] == 0;
        return clean * 2 >= (long long)(g.D - 2) * g.Wt;
    };
    FusedGrid fg;
    fg.nbx = g.Wt;
    // (the obstacle form of the kernel could take the floor's solid cells like any others and start at row 2,
    // but an empty tunnel would then run that form everywhere for one more row: 3 % slower)
    fg.y_first = 2;
    while (fg.y_first < std::min(g.H / 4, 8) && !row_clean(fg.y_first - 1)) fg.y_first++;
    fg.y_end = g.H - 2;
    while (fg.y_end > g.H - std::min(g.H / 4, 8) && !row_clean(fg.y_end)) fg.y_end--;
    if (fg.y_end - fg.y_first < 1) return LBM_OK;
    fg.zl_first = g.slab ? 2 : 3;                                   // widest possible range of second-step layers:
    fg.zl_end = g.slab ? g.D : g.D - 1;                             // a slab's outer layers are its boundary layers, a domain's its walls
    // With the WALL form of the kernel (k_fused2.h) the blocks reach the walls: every row that is not solid throughout
    // (the wind tunnel's floor plane is) and, in an undecomposed domain, every layer.  The wind tunnel's outlet tile
    // column stays inside the bounds above (yo_first .. zo_end: its blocks must not hold a wall cell).
    const bool walls = c->f2_walls && c->f2_obst;
    const int yo_first = fg.y_first, yo_end = fg.y_end;
    if (walls) {
        auto row_solid = [&](int y) {
            for (int z = 0; z < g.D; z++)
                for (int xs = 0; xs < g.Wt; xs++)
                    if (fl[tile(xs, y, z)] != 2) return false;
            return true;
        };
        fg.y_first = 0;
        while (fg.y_first < g.H / 4 && row_solid(fg.y_first)) fg.y_first++;
        fg.y_end = g.H;
        while (fg.y_end > g.H - g.H / 4 && row_solid(fg.y_end - 1)) fg.y_end--;
        if (!g.slab) { fg.zl_first = 1; fg.zl_end = g.D + 1; }
    }
    if (g.H >= (1 << 16) || g.D + 2 >= (1 << 14)) return LBM_OK;    // block list entry: 16 bits each for the first row and first layer, 14 for the layers

    // ---- the plan: y tiles, runs, blocks, launch order ------------------------------------------------------------
    // A tile takes at most TY rows and a CU holds one workgroup (132 KB of LDS), so a launch runs in rounds of `ncu`
    // blocks, and a part-filled last round costs as much as a full one.  Taller tiles and longer blocks are not
    // always better: 1024 x 512 x 64 (one of eight slabs of configs[3]) has 508 rows = 43 tiles of 12 = 688 tile
    // columns = 2.7 rounds on 256 CUs, while 47-48 tiles of 10-11 rows are three full rounds of cheaper layers; and
    // around an obstacle the number of blocks depends on where the runs break (766 blocks are three rounds, 770
    // four).  So the plan is SEARCHED: for every candidate number of y tiles (rows dealt evenly) the runs are found,
    // for every candidate piece length the blocks are cut, ordered and dealt to the XCDs, and the launch is played
    // through (every XCD hands its range of the list, in order, to whichever of its CUs is free) with a block
    // costing (layers + 2 lead-in layers) x (rows + row_cost: the waves that do first steps only, the barriers);
    // the candidate whose launch ends first wins.
    const int rows = fg.y_end - fg.y_first, layers_all = fg.zl_end - fg.zl_first;
    if (layers_all < 1) return LBM_OK;
    double row_cost = 4.0;   // (10 against 11 rows: 6 % less time per layer, tools/chain_spread.py)
    if (const char* t = getenv("LBM_FUSED_ROWCOST")) row_cost = atof(t);
    // (measured per block they cost 0.91 / 0.88 of an inner column's; planned with that, 256^3 goes from one round of 252
    // blocks to two of 252 and loses 8 %: blocks of different rounds do not march in step.  Left at 1 / 1)
    double edge_cost[2] = {1.0, 1.0};
    if (const char* t = getenv("LBM_FUSED_EDGECOST")) { edge_cost[0] = atof(t); if (const char* u = strchr(t, ',')) edge_cost[1] = atof(u + 1); else edge_cost[1] = edge_cost[0]; }
    constexpr int MIN_LAYERS = 6;   // shorter runs do not pay for their two lead-in layers: left to the rim
    struct YTile { int y0, n; };
    struct Run { int bx, by, y0, n, a, len; size_t dirty; };   // rows [y0, y0 + n) of y tile `by`; dirty: where its layers' obstacle counts start in Plan::dirty
    struct Block { unsigned bx, by, y0, n, zs, len, clean, plain; };   // clean: no solid cell within reach of its first steps; plain: no wall of the domain either
    struct Plan {
        std::vector<YTile> ytile;
        std::vector<Run> runs;
        std::vector<int> dirty;          // per run: running count of its first-step layers (a .. a + len + 1) that an obstacle reaches
        std::vector<Block> all;          // in launch order: eight ranges, one per XCD
        size_t range[9] = {0};
        bool obst_blocks = false, any_dirty = false;
        int zchunk = 0;
        double makespan = 1e300, covered = 0.0;
    };
    const bool obst = c->f2_obst;
    // LBM_FUSED_FORM (probes): 1 = the obstacle form of the kernel, 2 = its wall form, whatever the plan needs
    const int force_form = (c->probe && getenv("LBM_FUSED_FORM")) ? atoi(getenv("LBM_FUSED_FORM")) : 0;

    // rows dealt evenly, the taller tiles spread over the height (and so over the XCDs' ranges)
    // (`used` < rows: only the lowest `used` rows are tiled, the rest is left to the rim)
    auto make_tiles = [&](int ny, int used, Plan& pl) {
        pl.ytile.resize((size_t)ny);
        for (int i = 0, y = fg.y_first; i < ny; i++) {
            pl.ytile[(size_t)i] = {y, (int)((long long)used * (i + 1) / ny - (long long)used * i / ny)};
            y += pl.ytile[(size_t)i].n;
        }
    };
    // Runs: for every tile column (x tile, y tile) the maximal runs of layers a block can take.  Everything the
    // first step touches -- one more tile in x, one more row in y and one more layer in z on every side -- must
    // be free of the domain's walls (the kernel has no wall rows / layers; rows are kept off them by y_first /
    // y_end).  Obstacles: with f2_obst the kernel bounces back from solid cells itself, so flagged and all-solid
    // tiles may lie inside a block (but for the wind tunnel's outlet tile column, whose outlet branch may copy
    // from a solid cell's frozen values); without it every tile in reach must be an interior tile (flag 0).
    // A run of usable layers [a, b] carries second steps for the layers a+1 .. b-1; layers in which most of the
    // block is solid are not worth a pass and end a run as well.
    // (`wall_ok`: rows and layers outside the domain and wall tiles with nothing solid in reach, flag 3, count as clean)
    auto tiles_clean = [&](int bx, int y0, int tyn, int z_first, int z_last, bool wall_ok) {   // 0-based layers of first steps
        for (int z = std::max(z_first, wall_ok ? 0 : z_first); z <= std::min(z_last, wall_ok ? g.D - 1 : z_last); z++)
            for (int y = std::max(y0 - 1, wall_ok ? 0 : y0 - 1); y <= std::min(y0 + tyn, wall_ok ? g.H - 1 : y0 + tyn); y++)
                for (int xs = std::max(0, bx - 1); xs <= std::min(g.Wt - 1, bx + 1); xs++) {
                    const unsigned char f = fl[tile(xs, y, z)];
                    if (f != 0 && !(wall_ok && f == 3)) return false;
                }
        return true;
    };
    auto find_runs = [&](Plan& pl) {
        const int ny = (int)pl.ytile.size();
        std::vector<std::vector<Run>> runs_of((size_t)ny);
#pragma omp parallel for schedule(dynamic, 1)
        for (int by = 0; by < ny; by++) {
            for (int bx = 0; bx < fg.nbx; bx++) {
                // the wind tunnel's outlet lane copies from its own column, off the row / layer at a wall (quirk Q2) and
                // possibly from a solid cell's frozen values: the tile column that holds it -- and the one whose halo
                // column it is, when the last tile is one cell wide -- takes interior tiles only, trimmed to the rows
                // that keep off the walls
                const bool outlet_tile = c->preset == LBM_PRESET_WIND_TUNNEL_REF &&
                                         (bx == g.Wt - 1 || (walls && bx == g.Wt - 2 && (g.W - 1) % 64 == 0));
                int y0 = pl.ytile[(size_t)by].y0, tyn = pl.ytile[(size_t)by].n;
                if (walls && outlet_tile) {
                    const int ya = std::max(y0, yo_first), yb = std::min(y0 + tyn, yo_end);
                    if (yb - ya < std::min(4, tyn)) continue;          // a sliver: left to the rim
                    y0 = ya; tyn = yb - ya;
                }
                const bool wall_ok = walls && !outlet_tile;
                // (with walls in reach of the blocks, an undecomposed domain has a virtual usable layer either side)
                const int z_lo = (wall_ok && !g.slab) ? -1 : 0, z_hi = (wall_ok && !g.slab) ? g.D : g.D - 1;
                auto usable = [&](int z) {   // as a layer of first steps
                    if (z < 0 || z >= g.D) return true;
                    if (!obst || outlet_tile) return tiles_clean(bx, y0, tyn, z, z, false);
                    if (wall_ok) return true;
                    const int zg = z + g.z0;
                    return zg > 0 && zg < g.Dg - 1;
                };
                auto wanted = [&](int z) {   // as a layer of second steps
                    if (!obst || outlet_tile) return true;
                    int fluid = 0;
                    for (int y = y0; y < y0 + tyn; y++) fluid += fl[tile(bx, y, z)] != 2;
                    return fluid * 3 >= tyn;   // (thresholds from 1 % to 100 % of the rows move configs[2] by under 1 %)
                };
                int z = z_lo;
                while (z <= z_hi) {
                    if (!usable(z)) { z++; continue; }
                    const int a = z;
                    while (z <= z_hi && usable(z)) z++;
                    const int b = z - 1;                                 // usable layers [a, b] (0-based): second steps a+1 .. b-1
                    int s = a + 1;
                    while (s <= b - 1) {
                        if (!wanted(s)) { s++; continue; }
                        int e = s;
                        while (e + 1 <= b - 1 && wanted(e + 1)) e++;
                        if (e - s + 1 >= MIN_LAYERS) runs_of[(size_t)by].push_back({bx, by, y0, tyn, s - 1, e - s + 1, 0});
                        s = e + 1;
                    }
                }
            }
        }
        pl.runs.clear();
        for (auto& v : runs_of) pl.runs.insert(pl.runs.end(), v.begin(), v.end());
        // which first-step layers of a run an obstacle reaches: a block none of whose layers is reached reads no
        // neighbour words (k_fused2.h), and a launch without any such layer (and without walls) takes the plain form
        // of the kernel
        size_t total = 0;
        for (Run& r : pl.runs) { r.dirty = total; total += (size_t)r.len + 3; }
        pl.dirty.assign(total, 0);
        bool any = false;
        if (obst) {
#pragma omp parallel for schedule(dynamic, 16) reduction(|| : any)
            for (long long i = 0; i < (long long)pl.runs.size(); i++) {
                const Run& r = pl.runs[(size_t)i];
                int* d = pl.dirty.data() + r.dirty;
                for (int j = 0; j < r.len + 2; j++) d[j + 1] = d[j] + (tiles_clean(r.bx, r.y0, r.n, r.a + j, r.a + j, walls) ? 0 : 1);
                any = any || d[r.len + 2] > 0;
            }
        }
        pl.any_dirty = any;
        pl.obst_blocks = any || walls || force_form >= 1;
    };
    // blocks of at most `zchunk` layers, in launch order, and the launch played through
    auto block_cost = [&](const Plan& pl, const Block& k) {
        // (a tile column at an x end of the domain has one halo column less to compute and no neighbour's lines to share:
        // 256^3, three pieces per column, us per block: 497 / 546 / 546 / 480 for the four columns, tools/chain_spread.py)
        const double edge = (k.bx == 0 ? edge_cost[0] : 1.0) * ((int)k.bx == g.Wt - 1 ? edge_cost[1] : 1.0);
        return (double)(k.len + 2) * (k.n + row_cost) * edge * ((pl.obst_blocks && !k.clean) ? 1.06 : ((pl.obst_blocks && !k.plain) ? 1.025 : 1.0));
    };
    auto cut_blocks = [&](Plan& pl, int zchunk) {
        pl.zchunk = zchunk;
        pl.all.clear();
        pl.covered = 0.0;
        for (const Run& r : pl.runs) {
            const int pieces = (r.len + zchunk - 1) / zchunk;
            const int per = (r.len + pieces - 1) / pieces;
            for (int s0 = 0; s0 < r.len; s0 += per) {
                const int n = std::min(per, r.len - s0);
                const int zs = r.a + 1 + s0 + 1;                     // local layer index (ghost offset): z + 1
                const int* d = pl.dirty.data() + r.dirty;            // first steps: 0-based layers zs - 2 .. zs - 1 + n = run layers s0 .. s0 + n + 1
                const unsigned clean = pl.obst_blocks && d[s0 + n + 2] - d[s0] == 0 ? 1 : 0;
                // first steps: rows y0 - 1 .. y0 + n, local layers zs - 1 .. zs + n: none of them on a wall of the domain
                const bool off_walls = r.y0 - 1 >= 1 && r.y0 + r.n <= g.H - 2 && (zs - 2) + g.z0 >= 1 && (zs + n - 1) + g.z0 <= g.Dg - 2;
                pl.all.push_back({(unsigned)r.bx, (unsigned)r.by, (unsigned)r.y0, (unsigned)r.n, (unsigned)zs, (unsigned)n, clean,
                                  (unsigned)(clean && off_walls ? 1 : 0)});
                pl.covered += (double)n * r.n;
            }
        }
        // launch order: by layer range first, then y tile, then x tile, so that the workgroups that run together
        // (and the contiguous ranges the XCDs take, k_fused2.h) are neighbours in space
        std::stable_sort(pl.all.begin(), pl.all.end(), [&](const Block& p, const Block& q) {
            const unsigned cp = (p.zs - 1) / (unsigned)zchunk, cq = (q.zs - 1) / (unsigned)zchunk;
            if (cp != cq) return cp < cq;
            if (p.by != q.by) return p.by < q.by;
            return p.bx < q.bx;
        });
        // Eight contiguous ranges of the list, one per XCD, of equal cost; within a range the dear blocks go first so
        // that the cheap ones fill the tail.  Without obstacles all blocks of a tile height are alike and this is
        // the list cut into eight equal parts in its own order.
        double total = 0.0, done = 0.0;
        for (const Block& k : pl.all) total += block_cost(pl, k);
        size_t at = 0;
        const int per_xcd = std::max(1, ncu / 8);
        pl.makespan = 0.0;
        std::vector<double> free_at;
        for (int xcd = 0; xcd < 8; xcd++) {
            const size_t first = at;
            const double target = total * (xcd + 1) / 8;
            while (at < pl.all.size() && (xcd == 7 || done + block_cost(pl, pl.all[at]) / 2 <= target)) { done += block_cost(pl, pl.all[at]); at++; }
            std::stable_sort(pl.all.begin() + (long)first, pl.all.begin() + (long)at,
                             [&](const Block& p, const Block& q) { return block_cost(pl, p) > block_cost(pl, q); });
            pl.range[xcd] = first;
            pl.range[xcd + 1] = at;
            // the XCD's CUs take its blocks in list order, each when it is free (a min-heap of the times they are free at)
            free_at.assign((size_t)per_xcd, 0.0);
            for (size_t i = first; i < at; i++) {
                std::pop_heap(free_at.begin(), free_at.end(), std::greater<double>());
                free_at.back() += block_cost(pl, pl.all[i]);
                std::push_heap(free_at.begin(), free_at.end(), std::greater<double>());
            }
            pl.makespan = std::max(pl.makespan, *std::max_element(free_at.begin(), free_at.end()));
        }
    };

    Plan best;
    {
        const int ny_min = (rows + TY - 1) / TY;
        // candidates: (tiles, rows they cover).  Full tiles only, the remainder of the rows left to the rim, can be the
        // better deal: 256 rows are 21 tiles of 12 + 4 rows, and 4 x 21 tile columns in three pieces each are 252
        // blocks = one round on 256 CUs, where 22 tiles make 264
        std::vector<std::pair<int, int>> ny_candidates;
        if (c->f2_ny > 0) ny_candidates.push_back({std::min(rows, std::max(ny_min, c->f2_ny)), rows});
        else {
            if (rows % TY != 0 && rows / TY >= 1) ny_candidates.push_back({rows / TY, rows / TY * TY});
            for (int n = ny_min; n <= std::min(rows, ny_min + ny_min / 2 + 1); n++) ny_candidates.push_back({n, rows});
        }
        // what a candidate leaves to the rim costs two single steps there: about 2.5 x its share of a fused pass
        const double full = (double)rows * layers_all * fg.nbx;
        double best_score = 1e300;
        int best_ny = 0, best_zc = 0, best_used = 0;
        for (const auto& cand : ny_candidates) {
            const int ny = cand.first;
            Plan pl;
            make_tiles(ny, cand.second, pl);
            find_runs(pl);
            if (pl.runs.empty()) continue;
            int longest = 0;
            for (const Run& r : pl.runs) longest = std::max(longest, r.len);
            std::vector<int> zcs;
            if (c->f2_zsplit > 0) {
                zcs.push_back(std::min(max_chunk, std::max(MIN_LAYERS, (layers_all + c->f2_zsplit - 1) / c->f2_zsplit)));
            } else {
                for (int pieces = 1; pieces <= 16; pieces++) {
                    const int zc = std::min(max_chunk, (longest + pieces - 1) / pieces);
                    if (zc < MIN_LAYERS) break;
                    if (zcs.empty() || zcs.back() != zc) zcs.push_back(zc);
                }
            }
            for (int zc : zcs) {
                cut_blocks(pl, zc);
                const double score = pl.makespan + 2.5 * (full - pl.covered) * (TY + row_cost) / TY / ncu;
                if (const char* dbg = getenv("LBM_FUSED_DEBUG"))
                    if (dbg[0] == '3') fprintf(stderr, "[fused]   candidate: %d y tiles, pieces of %d layers: %zu blocks, launch %.0f, covered %.0f of %.0f, score %.0f\n",
                                               ny, zc, pl.all.size(), pl.makespan, pl.covered, full, score);
                if (score < best_score * (1.0 - 1e-9)) { best_score = score; best_ny = ny; best_zc = zc; best_used = cand.second; }
            }
        }
        if (best_ny == 0) return LBM_OK;
        make_tiles(best_ny, best_used, best);
        find_runs(best);
        cut_blocks(best, best_zc);
    }
    if (best.all.empty()) return LBM_OK;
    const std::vector<YTile>& ytile = best.ytile;
    const std::vector<Run>& runs = best.runs;
    const std::vector<Block>& all = best.all;
    const int zchunk = best.zchunk;
    fg.nby = (int)ytile.size();
    fg.zchunk = zchunk;
    c->f2_obst_blocks = best.obst_blocks;
    c->f2_wall_blocks = walls || force_form >= 2;

    std::vector<unsigned char> covered((size_t)ntiles, 0);
    long long fused_cells = 0;
    for (const Block& k : all) {
        const int y0 = (int)k.y0, tyn = (int)k.n;
        for (int zz = (int)k.zs - 1; zz < (int)(k.zs - 1 + k.len); zz++)
            for (int y = y0; y < y0 + tyn; y++) covered[tile((int)k.bx, y, zz)] = 1;
        fused_cells += (long long)std::min(64, g.W - (int)k.bx * 64) * tyn * k.len;
    }
    if (c->f2_obst_blocks) {   // fused_cells counts cells the kernel UPDATES: take the solid cells inside the blocks off
        std::vector<unsigned long long> mw((size_t)(g.D + 2) * g.H * g.Wm);
        HIPCHK(c, hipMemcpyAsync(mw.data(), c->mask, mw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        long long solid = 0;
#pragma omp parallel for schedule(static) reduction(+ : solid)
        for (int z = 0; z < g.D; z++)
            for (int y = 0; y < g.H; y++)
                for (int xs = 0; xs < g.Wt; xs++)
                    if (covered[tile(xs, y, z)] && fl[tile(xs, y, z)] != 0)
                        solid += __builtin_popcountll(mw[((size_t)(z + 1) * g.H + y) * g.Wm + xs]);
        fused_cells -= solid;
    }
    if (getenv("LBM_FUSED_DEBUG")) {   // planning summary for tools/fused_bench.py
        long long f0 = 0, f1 = 0, f2 = 0;
        for (long long i = 0; i < ntiles; i++) { f0 += fl[(size_t)i] == 0; f1 += fl[(size_t)i] == 1; f2 += fl[(size_t)i] == 2; }
        size_t nclean = 0, nplain = 0;
        double work = 0.0;
        for (const Block& k : all) { nclean += k.clean; nplain += k.plain; work += block_cost(best, k); }
        fprintf(stderr, "[fused] %dx%dx%d TY %d rows [%d,%d) in %d y tiles, layers [%d,%d) tiles: %lld interior %lld boundary %lld solid; %zu runs, zchunk %d, %zu blocks (%zu clean of obstacles, %zu of walls too), "
                "planned launch %.0f against %.0f for perfectly spread work, fused %.1f %% of cells\n",
                g.W, g.H, g.D, TY, fg.y_first, fg.y_end, fg.nby, fg.zl_first, fg.zl_end, f0, f1, f2, runs.size(), zchunk, all.size(), nclean, nplain,
                best.makespan, work / (8 * std::max(1, ncu / 8)), 100.0 * (double)fused_cells / (double)cells(c));
        std::vector<long long> per_by((size_t)fg.nby, 0);
        for (const Run& r : runs) per_by[(size_t)r.by] += r.len;
        if (getenv("LBM_FUSED_DEBUG")[0] == '2')
            for (int by = 0; by < fg.nby; by++)
                fprintf(stderr, "[fused]   y tile %d (rows %d..%d): %.1f %% of its column-layers\n", by, ytile[(size_t)by].y0, ytile[(size_t)by].y0 + ytile[(size_t)by].n - 1,
                        100.0 * (double)per_by[(size_t)by] / ((double)fg.nbx * (fg.zl_end - fg.zl_first)));
    }
    if (fused_cells * 4 < cells(c)) return LBM_OK;                  // under a quarter of the domain: not worth the rim's overhead
    std::vector<unsigned long long> blocks(9, 0ull);
    {
        size_t longest = 0;
        for (int xcd = 0; xcd < 8; xcd++) {
            blocks[(size_t)xcd] = best.range[xcd];
            blocks[(size_t)xcd + 1] = best.range[xcd + 1];
            longest = std::max(longest, best.range[xcd + 1] - best.range[xcd]);
        }
        fg.nblocks = (int)(8 * longest);
        c->f2_rounds = (fg.nblocks + ncu - 1) / ncu;
    }
    if (const char* dbg = getenv("LBM_FUSED_DEBUG"))
        if (dbg[0] == '4')   // the block list in launch order (tools/chain_spread.py --blocks joins it with the stamps)
            for (int xcd = 0; xcd < 8; xcd++)
                for (size_t i = best.range[xcd]; i < best.range[xcd + 1]; i++)
                    fprintf(stderr, "[fused-block] %zu %u %u %u %u %u %u\n", 8 * (i - best.range[xcd]) + (size_t)xcd, all[i].bx, all[i].y0, all[i].n, all[i].zs, all[i].len, all[i].clean);
    blocks.reserve(9 + all.size());
    for (const Block& k : all)
        blocks.push_back((unsigned long long)k.bx | ((unsigned long long)k.y0 << 12) | ((unsigned long long)k.n << 28) |
                         ((unsigned long long)k.zs << 32) | ((unsigned long long)k.len << 48) | ((unsigned long long)k.plain << 62) | ((unsigned long long)k.clean << 63));

    // rim = tiles the fused kernel does not cover (all-solid tiles need no update); its second step pulls
    // t+1 values from the 26 neighbouring tiles, so the first rim step covers the rim dilated by one tile
    std::vector<unsigned char> rim((size_t)ntiles), a((size_t)ntiles), b((size_t)ntiles);
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < ntiles; i++) rim[(size_t)i] = !covered[(size_t)i] && fl[(size_t)i] != 2;
#pragma omp parallel for schedule(static)
    for (long long rowi = 0; rowi < (long long)g.D * g.H; rowi++) {      // dilate along x
        const unsigned char* r = rim.data() + (size_t)rowi * g.Wt;
        unsigned char* o = a.data() + (size_t)rowi * g.Wt;
        for (int xs = 0; xs < g.Wt; xs++) o[xs] = r[xs] | (xs > 0 ? r[xs - 1] : 0) | (xs + 1 < g.Wt ? r[xs + 1] : 0);
    }
#pragma omp parallel for schedule(static)
    for (int z = 0; z < g.D; z++)                                        // along y
        for (int y = 0; y < g.H; y++)
            for (int xs = 0; xs < g.Wt; xs++)
                b[tile(xs, y, z)] = a[tile(xs, y, z)] | (y > 0 ? a[tile(xs, y - 1, z)] : 0) | (y + 1 < g.H ? a[tile(xs, y + 1, z)] : 0);
#pragma omp parallel for schedule(static)
    for (int z = 0; z < g.D; z++)                                        // along z
        for (int y = 0; y < g.H; y++)
            for (int xs = 0; xs < g.Wt; xs++)
                a[tile(xs, y, z)] = b[tile(xs, y, z)] | (z > 0 ? b[tile(xs, y, z - 1)] : 0) | (z + 1 < g.D ? b[tile(xs, y, z + 1)] : 0);
    std::vector<unsigned long long> list[2];
    long long rim_cells = 0, tail_cells = 0;
    for (int z = 0; z < g.D; z++)
        for (int y = 0; y < g.H; y++)
            for (int xs = 0; xs < g.Wt; xs++) {
                const size_t t = tile(xs, y, z);
                if (fl[t] == 2) continue;
                if (g.slab && (z == 0 || z == g.D - 1)) continue;   // a slab's boundary layers: lbm_pair_boundary_*()
                const unsigned long long e = (unsigned long long)xs | ((unsigned long long)y << 16) | ((unsigned long long)(z + 1) << 40);
                if (a[t]) { list[0].push_back(e); tail_cells += std::min(64, g.W - xs * 64); }
                if (rim[t]) { list[1].push_back(e); rim_cells += std::min(64, g.W - xs * 64); }
            }
    for (auto& l : list)
        while (l.size() % 4) l.push_back(~0ull);

    if (best.any_dirty) {   // the solid-neighbour words of every cell (k_fused2.h); without them the blocks must keep clear of obstacles
        const size_t words = (size_t)(g.D + 2) * g.H * g.Wp;
        if (!c->f2_nbw && hipMalloc(&c->f2_nbw, words * sizeof(unsigned)) != hipSuccess) {
            (void)hipGetLastError();
            c->f2_nbw = nullptr; c->f2_obst = false; c->fused_ready = false;
            return fused_prepare(c);
        }
        HIPCHK(c, hipMemsetAsync(c->f2_nbw, 0, words * sizeof(unsigned), c->stream));
        hipLaunchKernelGGL(k_f2_nb_words, grid1((long long)g.D * g.H * g.Wp), dim3(256), 0, c->stream, c->mask, c->f2_nbw, c->g);
        HIPCHK(c, hipGetLastError());
    }
    if (!c->ft) {  // third buffer; without it the fused path stays off
        if (hipMalloc(&c->ft, c->pop_bytes) != hipSuccess) { (void)hipGetLastError(); c->ft = nullptr; return LBM_OK; }
        hipLaunchKernelGGL(k_sync_solid, grid1(cells(c)), dim3(256), 0, c->stream, c->f[c->cur], c->ft, c->mask, c->g);
        HIPCHK(c, hipGetLastError());
    }
    int rc = fused_upload(c, (void**)&c->f2_blocks, &c->f2_blocks_cap, blocks.data(), blocks.size() * sizeof(unsigned long long));
    for (int i = 0; i < 2 && !rc; i++) {
        rc = fused_upload(c, (void**)&c->rim_list[i], &c->rim_cap[i], list[i].data(), list[i].size() * sizeof(unsigned long long));
        c->rim_n[i] = list[i].size();
    }
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));   // the host vectors go out of scope
    fg.wide = c->f2_wide == 1 ? 1 : 0;
    c->f2_wide_pending = c->f2_wide < 0 && f2_wide_ok(c->f2_stage, c->f2_nbuf);
    fused_tune_reset(c);
    c->fg = fg;
    c->fused_cells = fused_cells;
    c->rim_cells = rim_cells;
    c->tail_cells = tail_cells;
    c->fused_usable = true;
    return LBM_OK;
}

// one k_fused2 launch on the context's stream; between lbm_timer_begin and lbm_timer_end it is bracketed by HIP
// events of its own, so that the roofline figure of the dominant kernel does not include the rim's launches
// Narrow or wide first-step loads for the plain blocks (k_fused2.h WIDE)?  Neither wins everywhere (256^3 and 512^3 wind
// tunnel: wide +2 % / +4.5 %; 1024 x 512 x 512, thin slabs, obstacles: narrow +2-3 %; profiles/r03_tuning.md section 7), and
// a few launches timed back to back do not predict the steady state (the rim's launches in between, what the previous
// pair left in the caches; nor do alternating ones: each form then runs in the other's wake).  So of the first
// launches of real work after a plan the first 16 run narrow and untimed (the first launches of a run are slower whatever
// the form), then come 10 narrow, 10 wide and 10 narrow ones bracketed by events, and the form with the smaller mean (the
// first 4 of each group left out: a form needs a few launches in its own wake) is kept.  Shorter runs stay narrow.
constexpr int F2_TUNE_WARM = 16, F2_TUNE_GROUP = 10, F2_TUNE_SKIP = 4;
constexpr int F2_TUNE_LAUNCHES = F2_TUNE_WARM + 3 * F2_TUNE_GROUP;
static inline int f2_tune_form(int i) { return i >= F2_TUNE_WARM + F2_TUNE_GROUP && i < F2_TUNE_WARM + 2 * F2_TUNE_GROUP ? 1 : 0; }
static inline bool f2_tune_counts(int i) { return i >= F2_TUNE_WARM && (i - F2_TUNE_WARM) % F2_TUNE_GROUP >= F2_TUNE_SKIP; }
static void fused_tune_reset(lbm_ctx* c) {
    for (hipEvent_t e : c->f2_tune_ev) (void)hipEventDestroy(e);
    c->f2_tune_ev.clear();
    c->f2_tune_n = 0;
}
// before launch number f2_tune_n of the tuning phase: which form it takes; after the last one, the verdict
static int fused_tune_next(lbm_ctx* c) {
    if (c->f2_tune_n < F2_TUNE_LAUNCHES) {
        c->fg.wide = f2_tune_form(c->f2_tune_n);
        return LBM_OK;
    }
    c->f2_wide_pending = false;
    HIPCHK(c, hipEventSynchronize(c->f2_tune_ev.back()));
    double sum[2] = {0.0, 0.0};
    int cnt[2] = {0, 0};
    for (int i = 0; i < F2_TUNE_LAUNCHES; i++) {
        const int form = f2_tune_form(i);
        if (!f2_tune_counts(i)) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->f2_tune_ev[2 * (size_t)i], c->f2_tune_ev[2 * (size_t)i + 1]) == hipSuccess) { sum[form] += ms; cnt[form]++; }
    }
    for (int f = 0; f < 2; f++) c->f2_wide_ms[f] = cnt[f] ? (float)(sum[f] / cnt[f]) : 0.f;
    c->fg.wide = (cnt[0] && cnt[1] && c->f2_wide_ms[1] < 0.995f * c->f2_wide_ms[0]) ? 1 : 0;
    if (getenv("LBM_FUSED_DEBUG"))
        fprintf(stderr, "[fused] first-step loads timed over %d launches: narrow %.1f us, wide %.1f us -> %s\n", F2_TUNE_LAUNCHES,
                c->f2_wide_ms[0] * 1e3, c->f2_wide_ms[1] * 1e3, c->fg.wide ? "wide" : "narrow");
    fused_tune_reset(c);
    return LBM_OK;
}

static int launch_fused_timed(lbm_ctx* c, const float* in, float* out) {
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (c->f2_wide_pending) {
        const int rc = fused_tune_next(c);
        if (rc) return rc;
        if (c->f2_wide_pending) {
            if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess) {   // no events: no tuning
                (void)hipGetLastError();
                if (t0) (void)hipEventDestroy(t0);
                t0 = t1 = nullptr;
                c->f2_wide_pending = false; c->fg.wide = 0; fused_tune_reset(c);
            } else {
                c->f2_tune_ev.push_back(t0); c->f2_tune_ev.push_back(t1); c->f2_tune_n++;
            }
        }
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {
        if (c->f2_ev_used + 2 > c->f2_ev.size() && c->f2_ev.size() < 1024) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) { c->f2_ev.push_back(a); c->f2_ev.push_back(b); }
        }
        if (c->f2_ev_used + 2 <= c->f2_ev.size()) { e0 = c->f2_ev[c->f2_ev_used]; e1 = c->f2_ev[c->f2_ev_used + 1]; c->f2_ev_used += 2; }
    }
    if (e0) HIPCHK(c, hipEventRecord(e0, c->stream));
    if (t0) HIPCHK(c, hipEventRecord(t0, c->stream));
    hipError_t e = launch_fused(c, in, out);
    if (e != hipSuccess) return fail(c, LBM_ERR_HIP, "k_fused2 launch failed: %s", hipGetErrorString(e));
    if (t1) HIPCHK(c, hipEventRecord(t1, c->stream));
    if (e1) HIPCHK(c, hipEventRecord(e1, c->stream));
    c->fused_launches++;
    return LBM_OK;
}

// A step of the rim inside the fused launch (F2Rim in k_fused2.h): undecomposed contexts its first, z-slabs its second.
// Pays where the launch has a ragged tail -- several rounds of blocks with the narrow first-step loads (obstacles, thin
// slabs) -- or the rim is more than a few per cent of the cells.  With the wide loads every block issues the same requests
// and all end within a microsecond of each other, in one round or in eight: such a launch only gets longer by the rim's
// step (256^3 -0.6 %, 512^3 -0.9 %; profiles/r03_tuning.md section 13).  Not while the library is still timing the two
// forms of the loads against each other: both are timed without it.
static bool fused_tail_pays(const lbm_ctx* c) {
    if (!c->f2_tail_rim || c->f2_wide_pending) return false;
    return (c->f2_rounds > 1 && !c->fg.wide) || c->rim_cells * 100 > cells(c) * c->f2_tail_pct;
}
static bool fused_tail_rim(const lbm_ctx* c) { return !c->g.slab && fused_tail_pays(c); }
// ... on a z-slab: the rim's SECOND step (lbm_pair_interior)
static bool fused_tail_rim_slab(const lbm_ctx* c) { return c->g.slab && !c->pair_rim_stream && fused_tail_pays(c); }

// two time steps: t (f[cur]) -> t+2 (f[1-cur]).  `first`: the first pair of a batch.
// The rim's two launches touch no cell the fused kernel touches; where they run depends on how big the rim is (below).
// In the side-stream mode both streams meet again before the next pair (which reads everything this one wrote) and
// after the last one.
static int fused_pair(lbm_ctx* c, bool first) {
    const float* in = c->f[c->cur];
    float* out = c->f[1 - c->cur];
    hipStream_t rim_stream = c->stream;
    // A rim that is more than a few per cent of the domain (obstacles, small domains) should not wait behind the fused
    // kernel with three kernel boundaries per pair: its first step rides in the tail of the fused launch (k_fused2.h
    // F2Rim; 512x256x256 + car, rim 6 %: 52 480 -> 53 440 MLUPS; 400x100x100: 43 640 -> 45 120).  Around a large empty box
    // (rim 3 %) the small launches cost the same or less behind the fused kernel (256^3: 0.2993 ms per step against 0.3012
    // with the tail, 0.3165 with both on a side stream).  LBM_FUSED_TAIL_RIM=0: the earlier rule -- both rim launches on
    // a side stream beside the fused kernel when the rim is more than 15 %.
    const bool tail = fused_tail_rim(c);
    const bool overlap = c->f2_overlap && !tail && c->rim_cells * 100 > cells(c) * c->f2_overlap_pct;
    // (the rule can change between two pairs of a batch -- when the timing of the load forms ends -- so a pair that keeps
    // to the context's stream still has to wait for a rim the previous one left on the side stream)
    if (!overlap && c->f2_rim_pending) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_f2_rim, 0));
        c->f2_rim_pending = false;
    }
    if (overlap) {
        if (!c->stream_rim) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->stream_rim, hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_f2_main, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_f2_rim, hipEventDisableTiming));
        }
        rim_stream = c->stream_rim;
        if (!first && c->f2_rim_pending) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_f2_rim, 0));   // the previous pair's rim
        HIPCHK(c, hipEventRecord(c->ev_f2_main, c->stream));                       // ... and its fused launch, and all before
        HIPCHK(c, hipStreamWaitEvent(rim_stream, c->ev_f2_main, 0));
    }
    c->f2_tail_now = tail ? 1 : 0;
    int rc = launch_fused_timed(c, in, out);
    c->f2_tail_now = 0;
    if (rc) return rc;
    if (!tail) rc = launch_stream_collide<4>(c, in, c->ft, c->p, 0, 1, 0, nullptr, nullptr, 0, rim_stream, c->rim_list[0], c->rim_n[0]);
    if (!rc) rc = launch_stream_collide<4>(c, c->ft, out, c->p, 0, 1, 0, nullptr, nullptr, 0, rim_stream, c->rim_list[1], c->rim_n[1]);
    if (rc) return rc;
    if (overlap) HIPCHK(c, hipEventRecord(c->ev_f2_rim, rim_stream));
    c->f2_rim_pending = overlap;
    c->cur = 1 - c->cur;
    return LBM_OK;
}
// after the last pair of a batch: later work on the context's stream sees the rim's results
static int fused_join(lbm_ctx* c) {
    if (c->f2_rim_pending) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_f2_rim, 0));
    c->f2_rim_pending = false;
    return LBM_OK;
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
    } else if (c->scheme != LBM_SCHEME_AB) {
        return fail(c, LBM_ERR_STATE, "in-place schemes: rho/u of a new field window exist after the next lbm_step()");
    } else if (c->pair_last) {
        return fail(c, LBM_ERR_STATE, "the last update was a pair of steps (lbm_pair_*): rho/u exist again after the next single step");
    } else if (c->prev_touched) {
        // the fields of the last step were settled with the old mask and then dropped by a window change;
        // the buffer they came from has been re-frozen with the new mask since
        return fail(c, LBM_ERR_STATE, "the mask or the populations changed since the last step: rho/u of another field window exist after the next lbm_step()");
    } else if (c->ended_on_pair) {
        // lbm_step() ended on a pair: f[1-cur] is the state two steps back, untouched.  One ordinary step from it into
        // the rim's buffer gives the state the last step read; the gather from there is the one below.
        const int K = c->ab_chunk > 0 ? c->ab_chunk : c->g.D;
        for (int zb = 0; zb < c->g.D && !rc; zb += K)
            rc = launch_stream_collide<0>(c, c->f[1 - c->cur], c->ft, c->p_last, 1 + zb, std::min(K, c->g.D - zb));
        if (!rc) rc = launch_stream_collide<1>(c, c->ft, nullptr, c->p_last, 1 + c->g.fz0, c->g.fnz);
        if (rc) return rc;
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

// Waves a workgroup puts side by side in x (1, 2 or 4; the rest of its 4 waves stack in y).  Wide
// workgroups stream best when the row divides evenly (256^3: 6.16 TB/s with 4, 5.81 with 2, 5.45 with
// 1), but a row of Wt 64-cell segments is padded to a multiple of that width with waves that exit at
// once: the reference's default 400x100x100 domain (Wt = 7) runs at 5.15 / 5.30 / 5.8 TB/s with 4 / 2 /
// 1.  Pick the width with the least padded waves, weighted by those measured rates.
static int pick_segsx_log2(int Wt) {
    const double penalty[3] = {1.12, 1.06, 1.0};
    int best = 0;
    double best_cost = 1e30;
    for (int l = 0; l < 3; l++) {
        const int segsx = 1 << l;
        if (l > 0 && segsx > Wt) break;
        const double cost = (double)((Wt + segsx - 1) / segsx * segsx) * penalty[l];
        if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && l > best)) { best_cost = cost; best = l; }
    }
    return best;
}

// Layers per launch.  Measured on MI355X (profiles/r01_tuning.md, "layers per launch"): a step cut into
// launches of about 1 GB of algorithmic traffic runs 2-4 % faster than one launch over the whole
// domain (256^3: 6.09 -> 6.27 TB/s at 96 layers, 512x256x256: 5.75 -> 5.98 at 32-64, 1024x512x512:
// 5.96 -> 6.08 at 16-32), as long as a launch keeps >= 32 K workgroups or so
// (2048x1024x128 loses 1-2 % below 8 layers; 1504x1536x1484 in place is best at 2-4 layers).  Chunks are balanced over the depth.
static int auto_chunk(const Geo& g, int segsx_log2) {
    const int segsx = 1 << segsx_log2;
    const long long bpl = (long long)((g.Wt + segsx - 1) / segsx) * ((g.H + (4 / segsx) - 1) / (4 / segsx));  // workgroups per layer
    const long long layer_bytes = (long long)g.W * g.H * 152;
    long long k = std::max(((1ll << 30) + layer_bytes - 1) / layer_bytes, (32768 + bpl - 1) / bpl);
    if (k >= g.D) return g.D;
    const long long n = (g.D + k - 1) / k;
    return (int)((g.D + n - 1) / n);
}

// the caller asks for rho/u (read-back, curl, device pointers)
static int fields_for_caller(lbm_ctx* c) {
    c->fields_wanted = true;
    return materialise_fields(c);
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
        if (opts->struct_size < (int)offsetof(lbm_opts, slide_chunk) || opts->struct_size > (int)sizeof(lbm_opts))
            return fail(nullptr, LBM_ERR_INVALID, "lbm_create: lbm_opts.struct_size %d not understood", opts->struct_size);
        memcpy(&o, opts, (size_t)opts->struct_size);
    }
    if (o.preset != LBM_PRESET_WIND_TUNNEL_REF && o.preset != LBM_PRESET_CAVITY)
        return fail(nullptr, LBM_ERR_INVALID, "lbm_create: unknown preset %d", o.preset);
    if (o.scheme != LBM_SCHEME_AB && o.scheme != LBM_SCHEME_AA && o.scheme != LBM_SCHEME_SLIDE)
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
    const bool slide = o.scheme == LBM_SCHEME_SLIDE;
    if (slide) {
        if (o.slide_chunk < 0) { delete c; return fail(nullptr, LBM_ERR_INVALID, "lbm_create: slide_chunk %d", o.slide_chunk); }
        // default: the measured-best launch size, but at most a quarter of the depth (or 256 MB, whichever
        // is more) in spare layers -- the scheme exists to save memory; slide_chunk / LBM_SLIDE_CHUNK override
        const long long layer_pop_bytes = (long long)Q * width * height * (long long)sizeof(float);
        const int cap = (int)std::max<long long>(depth / 4, (256ll << 20) / layer_pop_bytes);
        c->slide_chunk = std::max(2, std::min(auto_chunk(g, pick_segsx_log2(g.Wt)), cap));
        if (o.slide_chunk > 0) c->slide_chunk = o.slide_chunk;
        if (const char* t = getenv("LBM_SLIDE_CHUNK")) c->slide_chunk = std::max(1, atoi(t));
        if (c->slide_chunk > depth) c->slide_chunk = depth;
        if (const char* t = getenv("LBM_SLIDE_SKEW")) c->slide_skew = std::max(0ll, atoll(t)) / 128 * 32;
    }
    const long long layers = (long long)depth + 2 + (slide ? c->slide_chunk + 2 : 0);  // SLIDE: room for the shifted copy
    long long guard;                                             // AA: floats in front of plane 0 that (y = -1, z = -1) reaches
    if (layout == LBM_LAYOUT_ROWS) {
        long long row_pad = 0;
        if (const char* t = getenv("LBM_ROW_PAD")) row_pad = atoll(t) / 32 * 32;
        g.plane = g.Wp;
        g.row = (long long)Q * g.Wp + (row_pad > 0 ? row_pad : 0);
        g.layer = rows_per_layer * g.row;
        guard = aa ? g.row + 32 : 0;
        c->pop_floats = layers * g.layer + c->slide_skew;
    } else {
        g.row = g.Wp;
        g.layer = rows_per_layer * g.Wp;
        long long skew_floats = (o.plane_skew_bytes >= 0 ? o.plane_skew_bytes : 3072) / 4;  // measured best of a 0..1 MiB sweep
        guard = aa ? g.Wp + 32 : 0;
        if (skew_floats < guard) skew_floats = guard;
        skew_floats = (skew_floats + 31) / 32 * 32;                                          // keep rows 128-B aligned
        g.plane = layers * g.layer + skew_floats + c->slide_skew;
        c->pop_floats = (long long)Q * g.plane;
    }
    c->guard_floats = guard;
    c->slide_shift = (long long)(c->slide_chunk + 2) * g.layer + c->slide_skew;
    c->segsx_log2 = pick_segsx_log2(g.Wt);
    if (const char* t = getenv("LBM_SEGSX")) c->segsx_log2 = std::min(2, std::max(0, atoi(t)));
    // measured on MI355X at 256^3 (profiles/r01_tuning.md): nt loads + sc1 (write-through) stores +
    // alternating z direction = +2.5 % over plain loads / nt stores.  LBM_TUNE overrides for A/B runs.
    // In place (AA) the write-through stores lose: partial-line neighbour writes no longer combine
    // in L2 (4.24 vs 5.39 TB/s), so AA keeps nt loads + nt stores + alternating z direction.
    // Row-interleaved A/B (tools/probes/tune_bench.py, 256^3): nt loads + nt stores 6.08 TB/s, with sc1 stores
    // 5.5, with alternating z 6.03 -> plain nt/nt is the default there.
    c->tune = o.scheme == LBM_SCHEME_AA ? 5 : (layout == LBM_LAYOUT_ROWS ? 1 : 7);
    if (const char* t = getenv("LBM_TUNE")) c->tune = atoi(t);
    c->probe = getenv("LBM_PROBE") != nullptr;
    {   // Workgroups go to the 8 XCDs round-robin in launch order.  When the workgroups per row divide 8
        // (W <= 2048 in powers of two) that hands every XCD a fixed x segment, which runs well; otherwise
        // (e.g. W = 1504: 6 per row) the assignment wanders, and giving each XCD whole rows is 2 % faster
        // for the pull kernel (5.77 -> 5.92 TB/s at 1504x1536; AA loses 1 %, tools/probes/xcd_rows_probe.py).
        const int segsx = 1 << c->segsx_log2, gx = (g.Wt + segsx - 1) / segsx;
        c->xcd_rows = o.scheme != LBM_SCHEME_AA && 8 % gx != 0;
    }
    if (const char* t = getenv("LBM_XCD_ROWS")) c->xcd_rows = atoi(t) != 0;
    c->ab_chunk = c->aa_chunk = auto_chunk(g, c->segsx_log2);
    if (const char* t = getenv("LBM_AB_CHUNK")) c->ab_chunk = c->aa_chunk = std::max(0, atoi(t));
    if (const char* t = getenv("LBM_LDS_PAD")) c->lds_pad = (unsigned)atoi(t);
    if (const char* t = getenv("LBM_PAIR_WAVE_BLOCKS")) c->pair_wave_blocks = atoi(t) != 0;
    c->fused_on = o.two_step != LBM_TWO_STEP_OFF;
    if (const char* t = getenv("LBM_FUSED")) c->fused_on = atoi(t) != 0;
    if (const char* t = getenv("LBM_FUSED_ZSPLIT")) c->f2_zsplit = std::max(0, atoi(t));
    if (const char* t = getenv("LBM_FUSED_OBST")) c->f2_obst = atoi(t) != 0;
    if (const char* t = getenv("LBM_FUSED_WALLS")) c->f2_walls = atoi(t) != 0;
    if (const char* t = getenv("LBM_FUSED_WIDE")) c->f2_wide = atoi(t) < 0 ? -1 : (atoi(t) != 0);
    if (const char* t = getenv("LBM_FUSED_NY")) c->f2_ny = std::max(0, atoi(t));
    if (const char* t = getenv("LBM_PAIR_RIM_STREAM")) c->pair_rim_stream = atoi(t) != 0;
    if (const char* t = getenv("LBM_SIDE_PRIO")) c->side_prio = atoi(t) != 0;
    if (const char* t = getenv("LBM_FUSED_OVERLAP")) c->f2_overlap = atoi(t) != 0;
    if (const char* t = getenv("LBM_FUSED_TAIL_RIM")) c->f2_tail_rim = atoi(t) != 0;
    if (const char* t = getenv("LBM_FUSED_TAIL_PCT")) c->f2_tail_pct = std::max(0, atoi(t));
    if (const char* t = getenv("LBM_FUSED_OVERLAP_PCT")) c->f2_overlap_pct = std::max(0, atoi(t));
    if (const char* t = getenv("LBM_FUSED_VARIANT")) {
        int ty = 0, st = 0, nb = 0;
        bool known = false;
        if (sscanf(t, "%d,%d,%d", &ty, &st, &nb) == 3)
            for (const F2Variant& v : kF2Variants) known = known || (v.ty == ty && v.stage == st && v.nbuf == nb);
        if (!known) { delete c; return fail(nullptr, LBM_ERR_INVALID, "lbm_create: LBM_FUSED_VARIANT=%s is not a compiled variant", t); }
        c->f2_ty = ty; c->f2_stage = st; c->f2_nbuf = nb;
    }
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
    CR(resolve_scratch(dev));
    if (o.boundary_stream && !o.stream) {
        fail(nullptr, LBM_ERR_INVALID, "lbm_create: lbm_opts.boundary_stream needs lbm_opts.stream (the caller orders the two streams)");
        return cleanup(LBM_ERR_INVALID);
    }
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
    } else if (slide) {
        CR(hipMalloc(&c->f[0], c->pop_bytes));
        CR(hipMalloc(&c->prev_mask, mask_bytes));
        CR(hipMemsetAsync(c->prev_mask, 0, mask_bytes, c->stream));
        if (!g.slab) {  // a slab reads the wrapped outlet column from its ghost layers like A/B
            const size_t ob_bytes = (size_t)OB_N * (depth + 2) * height * sizeof(float);
            CR(hipMalloc(&c->ob[0], ob_bytes));
            CR(hipMalloc(&c->ob[1], ob_bytes));
        }
    } else if (const char* t = getenv("LBM_AB_SINGLE")) {  // probe: both buffers in one allocation, a chosen distance apart
        const long long skew = std::max(0ll, atoll(t)) / 128 * 32;
        CR(hipMalloc(&c->alloc0, 2 * c->pop_bytes + (size_t)skew * sizeof(float)));
        c->f[0] = c->alloc0;
        c->f[1] = c->alloc0 + c->pop_floats + skew;
        c->f1_in_alloc0 = true;
    } else {
        CR(hipMalloc(&c->f[0], c->pop_bytes));
        CR(hipMalloc(&c->f[1], c->pop_bytes));
    }
    if (getenv("LBM_DEBUG_PTRS")) fprintf(stderr, "[lbm] f0=%p f1=%p pop_bytes=%zu\n", (void*)c->f[0], (void*)c->f[1], c->pop_bytes);
    CR(hipMalloc(&c->mask, mask_bytes));
    CR(hipMemsetAsync(c->mask, 0, mask_bytes, c->stream));  // d_solid.memset(0), lbm_solver.cu:415
    CR(hipMalloc(&c->flags, (size_t)depth * height * g.Wt));
    CR(hipEventCreate(&c->ev0));
    CR(hipEventCreate(&c->ev1));
    CR(hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
#undef CR
    *out = c;
    return LBM_OK;
}

int lbm_destroy(lbm_ctx* c) {
    if (!c) return LBM_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->alloc0) hipFree(c->alloc0); else hipFree(c->f[0]);
    if (!c->f1_in_alloc0) hipFree(c->f[1]);
    hipFree(c->mask); hipFree(c->flags);
    hipFree(c->ob[0]); hipFree(c->ob[1]); hipFree(c->prev_mask);
    hipFree(c->ft); hipFree(c->f2_nbw); hipFree(c->f2_blocks); hipFree(c->f2_stamps); hipFree(c->rim_list[0]); hipFree(c->rim_list[1]);
    for (hipEvent_t ev : c->f2_ev) hipEventDestroy(ev);
    fused_tune_reset(c);
    if (c->stream_rim) { hipStreamSynchronize(c->stream_rim); hipStreamDestroy(c->stream_rim); }
    if (c->ev_f2_main) hipEventDestroy(c->ev_f2_main);
    if (c->ev_f2_rim) hipEventDestroy(c->ev_f2_rim);
    if (c->ev_pair_b1) hipEventDestroy(c->ev_pair_b1);
    if (c->ev_pair_r1) hipEventDestroy(c->ev_pair_r1);
    hipFree(c->rho); hipFree(c->u); hipFree(c->curl); hipFree(c->solid_dense);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->ev_order) hipEventDestroy(c->ev_order);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
    return LBM_OK;
}

int lbm_init(lbm_ctx* c) {
    NEED(c);
    if (c->split_phase != 0 || c->pair_phase != 0) return fail(c, LBM_ERR_STATE, "%s: a split step or a pair of steps is in flight (lbm_step_finish / lbm_pair_finish missing)", "lbm_init");
    // the whole allocation: the AA scheme's ghost slots (y = -1, z = -1) lie in the skew region behind
    // the previous plane / in the front guard
    const long long first = -c->guard_floats, last = c->pop_floats;
    hipLaunchKernelGGL(k_fill_planes, grid1(last - first), dim3(256), 0, c->stream, c->f[0],
                       c->f[1] ? c->f[1] : c->f[0], c->g, c->layout == LBM_LAYOUT_ROWS ? 1 : 0, first, last);
    HIPCHK(c, hipGetLastError());
    if (c->scheme != LBM_SCHEME_AB) {
        if (c->ob[0]) {
            const long long per = (long long)(c->g.D + 2) * c->g.H;
            hipLaunchKernelGGL(k_ob_fill, grid1(per * OB_N), dim3(256), 0, c->stream, c->ob[0], c->ob[1], per);
            HIPCHK(c, hipGetLastError());
        }
        c->aa_phase = 0;  // w_k = w_inv(k): the equilibrium fill is valid in the streamed layout
        c->slide_off = 0;
        c->slide_pristine = c->scheme == LBM_SCHEME_SLIDE;
        c->slide_stepped = false;
    }
    if (c->ft) c->mask_dirty = true;  // the third buffer's frozen solid cells must follow (k_sync_solid)
    c->fields = FIELDS_INIT;  // rho = 1, u = 0 everywhere, solid cells included (lbm_solver.cu:62-63)
    c->prev_touched = false;
    c->pair_last = false;
    c->ended_on_pair = false;
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
    // nothing to settle without a field window; in place (AA, SLIDE) stale fields only arise from a moved
    // window and are refilled by the next lbm_step()
    if (c->fields == FIELDS_STALE && c->g.fnz > 0 && c->scheme == LBM_SCHEME_AB) {
        int rc = refresh_mask_state(c);
        if (rc) return rc;
        return materialise_fields(c);
    }
    return LBM_OK;
}

int lbm_add_solid_rect(lbm_ctx* c, int x, int y, int z, int w, int h, int d) {
    NEED(c);
    if (c->split_phase != 0 || c->pair_phase != 0) return fail(c, LBM_ERR_STATE, "%s: a split step or a pair of steps is in flight (lbm_step_finish / lbm_pair_finish missing)", "lbm_add_solid_rect");
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
    if (c->split_phase != 0 || c->pair_phase != 0) return fail(c, LBM_ERR_STATE, "%s: a split step or a pair of steps is in flight (lbm_step_finish / lbm_pair_finish missing)", "lbm_set_solid_u8");
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

// the mask from a dense uint8 array that already lives on the device (cuda_buffer::copy_from_device)
int lbm_set_solid_u8_device(lbm_ctx* c, const uint8_t* d_mask, size_t n) {
    NEED(c);
    if (c->split_phase != 0 || c->pair_phase != 0) return fail(c, LBM_ERR_STATE, "%s: a split step or a pair of steps is in flight (lbm_step_finish / lbm_pair_finish missing)", "lbm_set_solid_u8_device");
    if (!d_mask) return fail(c, LBM_ERR_INVALID, "lbm_set_solid_u8_device: mask is null");
    if (n != (size_t)cells(c)) return fail(c, LBM_ERR_INVALID, "lbm_set_solid_u8_device: expected %lld cells, got %zu", cells(c), n);
    int rc = before_mask_change(c);
    if (rc) return rc;
    const long long nwords = (long long)c->g.D * c->g.H * c->g.Wm;
    hipLaunchKernelGGL(k_pack_mask, grid1(nwords), dim3(256), 0, c->stream, d_mask, c->mask, c->g);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));   // the caller may reuse its array at once (cudaMemcpy semantics)
    c->mask_dirty = true;
    return LBM_OK;
}

// cudaMemcpy(..., cudaMemcpyDeviceToDevice) for hosts that do not link the HIP runtime themselves
int lbm_device_copy(lbm_ctx* c, void* d_dst, const void* d_src, size_t bytes) {
    NEED(c);
    if (!d_dst || !d_src) return fail(c, LBM_ERR_INVALID, "lbm_device_copy: null pointer");
    if (bytes == 0) return LBM_OK;
    HIPCHK(c, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LBM_OK;
}

int lbm_load_mesh(lbm_ctx* c, const char* obj_path) {
    NEED(c);
    if (c->split_phase != 0 || c->pair_phase != 0) return fail(c, LBM_ERR_STATE, "%s: a split step or a pair of steps is in flight (lbm_step_finish / lbm_pair_finish missing)", "lbm_load_mesh");
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
    if (c->split_phase != 0 || c->pair_phase != 0) return fail(c, LBM_ERR_STATE, "lbm_step: a split step is in flight (lbm_step_finish / lbm_pair_finish missing)");
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
    if (c->scheme == LBM_SCHEME_SLIDE) {
        rc = ensure_field_buffers(c);
        for (int s = 0; s < nsteps && !rc; s++) rc = slide_step(c, s == nsteps - 1 && c->g.fnz > 0);
        if (rc) return rc;
        c->p_last = c->p;
        c->steps += (uint64_t)nsteps;
        c->fields = FIELDS_VALID;
        c->curl_valid = false;
        return LBM_OK;
    }
    // rho/u are made on demand from the previous buffer (96 B/cell).  A caller that read them after its
    // previous lbm_step() call will most likely do so again (the engine's renderer does, every frame):
    // then the last step of this call writes them as it goes (20 B/cell).
    const bool fuse = c->fields_wanted && c->g.fnz > 0;
    if (fuse) {
        rc = ensure_field_buffers(c);
        if (rc) return rc;
    }
    int s = 0;
    // Two steps per pass wherever the batch allows it.  When the caller wants rho/u written (fuse) or the count is
    // odd, the LAST step of a call runs alone: rho/u of that step are recomputed on demand from the buffer it read,
    // which a fused pair would skip over.  An even count without fields runs in pairs to the end; should rho/u be
    // asked for after all, materialise_fields() takes the step in between once more.
    const bool all_pairs = !fuse && nsteps % 2 == 0;
    bool on_pair = false;
    if ((nsteps >= 3 || all_pairs) && fused_applies(c)) {
        rc = fused_prepare(c);
        if (rc) return rc;
        if (c->fused_usable) {
            on_pair = all_pairs;
            for (; s + (all_pairs ? 1 : 2) < nsteps; s += 2) {
                rc = fused_pair(c, s == 0);
                if (rc) return rc;
            }
            rc = fused_join(c);
            if (rc) return rc;
        }
    }
    for (; s < nsteps; s++) {
        rc = launch_ab_layers(c, 1, c->g.D, fuse && s == nsteps - 1);
        if (rc) return rc;
        c->cur = 1 - c->cur;  // std::swap(d_f, d_f_new), lbm_solver.cu:490
        c->zrev ^= 1;
    }
    c->p_last = c->p;
    c->steps += (uint64_t)nsteps;
    c->fields = fuse ? FIELDS_VALID : FIELDS_STALE;
    c->prev_touched = false;
    c->pair_last = false;
    c->ended_on_pair = on_pair;
    c->fields_wanted = false;
    c->curl_valid = false;
    return LBM_OK;
}

int launch_curl(lbm_ctx* c) {
    // layers per workgroup: as many as leave two workgroups for every CU (the chunk's two end layers are read twice)
    const int gx = (c->g.W + 63) / 64, gy = (c->g.H + CURL_NW * CURL_R - 1) / (CURL_NW * CURL_R);
    int zchunk = 64;
    while (zchunk > 8 && gx * gy * ((c->g.D + zchunk - 1) / zchunk) < 512) zchunk /= 2;   // (256 CUs)
    if (c->probe) { const char* t = getenv("LBM_CURL_ZCHUNK"); if (t && atoi(t) > 0) zchunk = atoi(t); }
    const int gz = (c->g.D + zchunk - 1) / zchunk;
    hipLaunchKernelGGL((k_curl<CURL_NW, CURL_R, CURL_U>), dim3(gx * gy * gz), dim3(CURL_NW * 64), 0, c->stream,
                       c->u, c->curl, c->g.W, c->g.H, c->g.D, zchunk);
    HIPCHK(c, hipGetLastError());
    return LBM_OK;
}

// The reference's frame loop on the device: every step() writes rho and u (lbm_solver.cu:336-337) and runs the curl pass
// (:488 -> :612-661).  `nframes` such steps back to back on the context's stream with no host round trip between
// them; afterwards rho / u (and curl with want_curl) are those of the last frame, exactly as after nframes calls of
// the reference's step().  172 B per cell and frame, 192 B with the curl pass.
int lbm_step_frames(lbm_ctx* c, int nframes, int want_curl) {
    NEED(c);
    if (nframes < 0) return fail(c, LBM_ERR_INVALID, "lbm_step_frames: nframes %d", nframes);
    if (nframes == 0) return LBM_OK;
    if (c->split_phase != 0 || c->pair_phase != 0) return fail(c, LBM_ERR_STATE, "lbm_step_frames: a split step is in flight");
    if (c->g.fnz == 0) return fail(c, LBM_ERR_STATE, "lbm_step_frames: the field window is empty (lbm_set_field_window)");
    if (want_curl && c->g.fnz != c->g.D)
        return fail(c, LBM_ERR_STATE, "lbm_step_frames: the curl's z differences need the whole domain in the field window");
    int rc = refresh_mask_state(c);
    if (!rc) rc = ensure_field_buffers(c);
    if (rc) return rc;
    if (want_curl && !c->curl) HIPCHK(c, hipMalloc(&c->curl, cells(c) * sizeof(float)));
    for (int s = 0; s < nframes && !rc; s++) {
        if (c->scheme == LBM_SCHEME_AA) rc = launch_aa_step(c, 1);
        else if (c->scheme == LBM_SCHEME_SLIDE) rc = slide_step(c, 1);
        else {
            rc = launch_ab_layers(c, 1, c->g.D, want_curl ? 1 : 5);
            c->cur = 1 - c->cur;
            c->zrev ^= 1;
        }
        if (!rc && want_curl) {
            rc = launch_curl(c);
            if (rc) return rc;
        }
    }
    if (rc) return rc;
    c->p_last = c->p;
    c->steps += (uint64_t)nframes;
    c->fields = FIELDS_VALID;
    c->prev_touched = false;
    c->pair_last = false;
    c->ended_on_pair = false;
    c->fields_wanted = true;     // this caller reads the fields every frame
    c->curl_valid = want_curl != 0;
    return LBM_OK;
}

int lbm_compute_curl(lbm_ctx* c) {
    NEED(c);
    int rc = refresh_mask_state(c);
    if (rc) return rc;
    rc = fields_for_caller(c);
    if (rc) return rc;
    if (c->curl_valid) return LBM_OK;
    if (c->g.fnz != c->g.D)
        return fail(c, LBM_ERR_STATE, "lbm_compute_curl: the curl's z differences need the whole domain in the field window");
    if (!c->curl) HIPCHK(c, hipMalloc(&c->curl, cells(c) * sizeof(float)));
    rc = launch_curl(c);
    if (rc) return rc;
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
    if (c->split_phase != 0 || c->pair_phase != 0) return fail(c, LBM_ERR_STATE, "lbm_set_streams: a split step is in flight");
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
    if (!rc) rc = fields_for_caller(c);
    if (rc) return rc;
    return read_back(c, rho, c->rho, n * sizeof(float));
}

int lbm_read_velocity(lbm_ctx* c, float* u, size_t n) {
    NEED(c);
    if (!u || n != (size_t)field_cells(c)) return fail(c, LBM_ERR_INVALID, "lbm_read_velocity: need %lld x 4 floats", field_cells(c));
    int rc = refresh_mask_state(c);
    if (!rc) rc = fields_for_caller(c);
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
    if (!rc) rc = fields_for_caller(c);
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

// The same sample without any allocation: the kernel writes into this device's copy of `scratch`.
int lbm_read_velocity_strided_noalloc(lbm_ctx* c, int sx, int sy, int sz, float* u, size_t n) {
    NEED(c);
    if (sx < 1 || sy < 1 || sz < 1) return fail(c, LBM_ERR_INVALID, "lbm_read_velocity_strided_noalloc: strides must be positive");
    const int nx = (c->g.W + sx - 1) / sx, ny = (c->g.H + sy - 1) / sy, nz = (c->g.fnz + sz - 1) / sz;
    const size_t need = (size_t)nx * ny * nz;
    if (!u || n != need) return fail(c, LBM_ERR_INVALID, "lbm_read_velocity_strided_noalloc: need %zu x 4 floats (%d x %d x %d samples)", need, nx, ny, nz);
    if (need > (size_t)LBM_NOALLOC_MAX_SAMPLES)
        return fail(c, LBM_ERR_INVALID, "lbm_read_velocity_strided_noalloc: %zu samples exceed the fixed scratch of %d (LBM_NOALLOC_MAX_SAMPLES); "
                    "use larger strides or lbm_read_velocity_strided", need, (int)LBM_NOALLOC_MAX_SAMPLES);
    if (need == 0) return LBM_OK;
    // rho/u are allocated on first use; this call may not do that
    if (!c->rho || !c->u)
        return fail(c, LBM_ERR_STATE, "lbm_read_velocity_strided_noalloc: the field buffers do not exist yet and this call allocates nothing: "
                    "read a field once (lbm_read_velocity, lbm_device_fields, ...) after lbm_create / lbm_set_field_window");
    float4* d = reinterpret_cast<float4*>(g_scratch_ptr[c->device]);
    if (!d) return fail(c, LBM_ERR_STATE, "lbm_read_velocity_strided_noalloc: no scratch address for device %d", c->device);
    int rc = refresh_mask_state(c);
    if (!rc) rc = fields_for_caller(c);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sample_velocity, grid1((long long)need), dim3(256), 0, c->stream, c->u, d, c->g.W, c->g.H, nx, ny, nz, sx, sy, sz);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(u, d, need * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LBM_OK;
}

int lbm_read_speed(lbm_ctx* c, float* speed, size_t n) {
    NEED(c);
    if (!speed || n != (size_t)field_cells(c)) return fail(c, LBM_ERR_INVALID, "lbm_read_speed: need %lld floats", field_cells(c));
    int rc = refresh_mask_state(c);
    if (!rc) rc = fields_for_caller(c);
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
        float* dev = current_populations(c) + (long long)k * g.plane + g.layer;  // first owned layer
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
    if (c->scheme == LBM_SCHEME_SLIDE) {  // pristine mode: the solid cells' slots were not carried along
        int rc = slide_fill_solids(c, c->prev_mask);
        if (rc) return rc;
    }
    return copy_populations(c, f, true);
}

int lbm_write_populations(lbm_ctx* c, const float* f, size_t n19) {
    NEED(c);
    if (c->split_phase != 0 || c->pair_phase != 0) return fail(c, LBM_ERR_STATE, "%s: a split step or a pair of steps is in flight (lbm_step_finish / lbm_pair_finish missing)", "lbm_write_populations");
    if (!f || n19 != (size_t)cells(c) * Q) return fail(c, LBM_ERR_INVALID, "lbm_write_populations: need %lld floats", cells(c) * Q);
    if (c->scheme == LBM_SCHEME_AA) return aa_copy_populations(c, const_cast<float*>(f), false);
    int rc = LBM_OK;
    rc = before_mask_change(c);  // settle rho/u of the last step before touching solid cells
    if (rc) return rc;
    rc = copy_populations(c, const_cast<float*>(f), false);
    if (rc) return rc;
    if (c->scheme == LBM_SCHEME_SLIDE) {  // the outlet column's side array follows the new populations
        c->slide_pristine = false;        // solid cells now hold what the caller wrote: carry them along
        if (c->ob[0]) {
            hipLaunchKernelGGL(k_ob_gather, grid1((long long)c->g.D * c->g.H), dim3(256), 0, c->stream,
                               current_populations(c), c->ob[0], c->ob[1], c->g);
            HIPCHK(c, hipGetLastError());
        }
        return LBM_OK;
    }
    c->mask_dirty = true;  // solid cells of the other buffer must follow
    return LBM_OK;
}

int lbm_device_fields(lbm_ctx* c, float** d_rho, float** d_u, float** d_curl, uint8_t** d_solid) {
    NEED(c);
    int rc;
    if (d_curl && c->g.fnz == c->g.D) {
        rc = lbm_compute_curl(c);
    } else {  // no curl asked for, or a partial field window (rho/u of the window only, no curl)
        rc = refresh_mask_state(c);
        if (!rc) rc = fields_for_caller(c);
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
    if (d_curl) *d_curl = c->g.fnz == c->g.D ? c->curl : nullptr;
    return LBM_OK;
}

// cuda_buffer<T>::copy_from_host / copy_from_device / memset on d_rho, d_u, d_curl (cuda_buffer.cuh:66-77,86-91).  In the
// reference these arrays are OUTPUTS of step() (lbm_solver.cu:336-337, :655): what a caller writes into them is what
// readers see until the next step rewrites them, and it never reaches the populations.  Same here: the fields are
// brought up to date, the first `count` elements overwritten, and the next lbm_step() makes them stale again.
int lbm_write_field(lbm_ctx* c, int field, const void* src, size_t count, int src_on_device) {
    NEED(c);
    if (!src) return fail(c, LBM_ERR_INVALID, "lbm_write_field: source is null");
    if (field < LBM_FIELD_DENSITY || field > LBM_FIELD_CURL) return fail(c, LBM_ERR_INVALID, "lbm_write_field: unknown field %d", field);
    const size_t n = (size_t)(field == LBM_FIELD_CURL ? cells(c) : field_cells(c));
    if (count > n) return fail(c, LBM_ERR_INVALID, "lbm_write_field: %zu elements, the field holds %zu", count, n);
    int rc = field == LBM_FIELD_CURL ? lbm_compute_curl(c) : refresh_mask_state(c);
    if (!rc && field != LBM_FIELD_CURL) rc = fields_for_caller(c);
    if (rc) return rc;
    void* dst = field == LBM_FIELD_DENSITY ? (void*)c->rho : (field == LBM_FIELD_VELOCITY ? (void*)c->u : (void*)c->curl);
    const size_t bytes = count * (field == LBM_FIELD_VELOCITY ? sizeof(float4) : sizeof(float));
    if (bytes) HIPCHK(c, hipMemcpyAsync(dst, src, bytes, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (field == LBM_FIELD_VELOCITY) c->curl_valid = false;   // a curl asked for now is the curl of what was written
    return LBM_OK;
}

// cudaMemcpy(..., cudaMemcpyDeviceToHost) for hosts that do not link the HIP runtime themselves: a prefix of a field
// whose device pointer lbm_device_fields returned (cuda_buffer<T>::copy_to_host with size < the field's)
int lbm_copy_to_host(lbm_ctx* c, void* dst, const void* d_src, size_t bytes) {
    NEED(c);
    if (!dst || !d_src) return fail(c, LBM_ERR_INVALID, "lbm_copy_to_host: null pointer");
    if (bytes == 0) return LBM_OK;
    return read_back(c, dst, d_src, bytes);
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
    out->population_bytes = (c->scheme == LBM_SCHEME_AB ? 2ull : 1ull) * c->pop_bytes;
    out->last_timed_ms = c->last_ms;
    out->last_timed_steps = c->last_steps;
    out->last_timed_launches = c->last_launches;
    out->population_bytes += (c->ft && c->fused_usable) ? c->pop_bytes : 0;   // the third buffer, while the fused path uses it
    out->fused_cells = (uint64_t)(c->fused_usable ? c->fused_cells : 0);
    out->rim_cells = (uint64_t)(c->fused_usable ? c->rim_cells : 0);
    out->last_timed_fused_launches = c->last_fused_launches;
    out->last_timed_fused_ms = c->last_fused_ms;
    out->fused_wide_loads = !c->fused_usable ? -1 : (c->f2_wide_pending ? -1 : c->fg.wide);
    out->fused_tuned_ms[0] = c->f2_wide_ms[0];
    out->fused_tuned_ms[1] = c->f2_wide_ms[1];
    out->fused_tail_cells = (uint64_t)(!c->fused_usable ? 0 : fused_tail_rim(c) ? c->tail_cells : fused_tail_rim_slab(c) ? c->rim_cells : 0);
    return LBM_OK;
}

int lbm_timer_begin(lbm_ctx* c) {
    NEED(c);
    int rc = refresh_mask_state(c);  // keep one-off set-up kernels out of the timed region
    if (rc) return rc;
    c->steps_at_begin = c->steps;
    c->launches_at_begin = c->launches;
    c->fused_launches_at_begin = c->fused_launches;
    c->f2_ev_used = 0;
    c->timing = true;
    // keep the one-off host work of the fused path out of the timed region as well -- for callers that use it: a
    // context that has never launched the fused kernel (single-step frame loops) prepares nothing here (the third
    // buffer is +50 % population memory)
    if ((fused_applies(c) || fused_applies_slab(c)) && c->fused_launches > 0) {
        rc = fused_prepare(c);
        if (rc) return rc;
    }
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
    c->last_launches = c->launches - c->launches_at_begin;
    c->last_fused_launches = c->fused_launches - c->fused_launches_at_begin;
    c->timing = false;
    double fms = 0.0;
    for (size_t i = 0; i + 1 < c->f2_ev_used; i += 2) {
        float t = 0.0f;
        if (hipEventElapsedTime(&t, c->f2_ev[i], c->f2_ev[i + 1]) == hipSuccess) fms += t;
    }
    // launches beyond the event pool are priced at the mean of the measured ones
    const size_t measured = c->f2_ev_used / 2;
    c->last_fused_ms = measured ? fms * (double)c->last_fused_launches / (double)measured : 0.0;
    return LBM_OK;
}

// Diagnostic: ONE k_fused2 launch from the current state into the other buffer with every workgroup writing its start
// and end clock (s_memrealtime, 100 MHz): how evenly the chains of fused_prepare() end.  The state does not advance;
// the buffer the launch writes is scratch between steps, but rho/u on demand of the last step are gone.
int lbm_fused_chain_stamps(lbm_ctx* c, uint64_t* stamps, size_t n, int* nworkgroups) {
    NEED(c);
    if (!nworkgroups) return fail(c, LBM_ERR_INVALID, "lbm_fused_chain_stamps: null argument");
    *nworkgroups = 0;
    if (c->split_phase != 0 || c->pair_phase != 0) return fail(c, LBM_ERR_STATE, "lbm_fused_chain_stamps: a split step is in flight");
    if (!fused_applies(c) && !fused_applies_slab(c)) return LBM_OK;
    int rc = refresh_mask_state(c);
    if (!rc) rc = fused_prepare(c);
    if (rc) return rc;
    if (!c->fused_usable) return LBM_OK;
    *nworkgroups = c->fg.nblocks;
    if (!stamps) return LBM_OK;                                  // size query
    if (n < 2 * (size_t)c->fg.nblocks) return fail(c, LBM_ERR_INVALID, "lbm_fused_chain_stamps: need %d x 2 stamps", c->fg.nblocks);
    const size_t bytes = 2 * (size_t)c->fg.nblocks * sizeof(unsigned long long);
    HIPCHK(c, hipMalloc(&c->f2_stamps, bytes));
    hipError_t e = hipMemsetAsync(c->f2_stamps, 0, bytes, c->stream);
    if (e == hipSuccess) e = launch_fused(c, c->f[c->cur], c->f[1 - c->cur]);
    if (e == hipSuccess) e = hipMemcpyAsync(stamps, c->f2_stamps, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(c->f2_stamps);
    c->f2_stamps = nullptr;
    c->prev_touched = true;
    if (e != hipSuccess) return fail(c, LBM_ERR_HIP, "lbm_fused_chain_stamps: %s", hipGetErrorString(e));
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
    if (c->pair_phase != 0) return fail(c, LBM_ERR_STATE, "%s: a pair of steps is in flight: its messages are written by lbm_pair_boundary_* and routed by lbm_halo_unpack_both", "lbm_halo_pack");
    int rc = halo_args_ok(c, side, d_message, "lbm_halo_pack");
    if (rc) return rc;
    if (c->scheme == LBM_SCHEME_AA) return aa_halo(c, side, d_message, true);
    const long long per = c->g.mlayer > c->g.H + 2 ? c->g.mlayer : c->g.H + 2;
    if (c->scheme == LBM_SCHEME_SLIDE && c->split_phase == 0) {  // pristine mode: solid cells of the boundary layers
        rc = slide_fill_solids(c, c->prev_mask);
        if (rc) return rc;
    }
    // after lbm_step_interior_layers the new state is still f[1-cur] until lbm_step_finish
    const float* src = halo_lattice(c);
    hipLaunchKernelGGL(k_halo_pack, dim3(blocks_for(per, 256), 6), dim3(256), 0, c->stream, src, d_message, c->g, side);
    HIPCHK(c, hipGetLastError());
    return LBM_OK;
}

int lbm_halo_unpack(lbm_ctx* c, int side, const float* d_message) {
    NEED(c);
    if (c->pair_phase != 0) return fail(c, LBM_ERR_STATE, "%s: a pair of steps is in flight: its messages are written by lbm_pair_boundary_* and routed by lbm_halo_unpack_both", "lbm_halo_unpack");
    int rc = halo_args_ok(c, side, d_message, "lbm_halo_unpack");
    if (rc) return rc;
    if (c->scheme == LBM_SCHEME_AA) return aa_halo(c, side, const_cast<float*>(d_message), false);
    if (c->scheme == LBM_SCHEME_SLIDE && c->split_phase == 1)
        return fail(c, LBM_ERR_STATE, "lbm_halo_unpack: sliding scheme: the ghost layers of the new lattice position are free only after lbm_step_interior_layers");
    const long long per = c->g.mlayer > c->g.H + 2 ? c->g.mlayer : c->g.H + 2;
    float* dst = halo_lattice(c);
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
    if (c->split_phase != 0 || c->pair_phase != 0) return fail(c, LBM_ERR_STATE, "%s: a split step or a pair of steps is in flight (lbm_step_finish / lbm_pair_finish missing)", "lbm_halo_set_mask_ghost");
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
    if (c->split_phase != 0 || c->pair_phase != 0) return fail(c, LBM_ERR_STATE, "lbm_step_boundary_layers: previous split step not finished");
    if ((d_msg_bottom || d_msg_top) && !c->g.slab)
        return fail(c, LBM_ERR_STATE, "lbm_step_boundary_layers: messages need a slab context");
    const bool was_dirty = c->mask_dirty;
    int rc = refresh_mask_state(c);
    if (rc) return rc;
    if (was_dirty && c->stream_b) {  // the set-up kernels ran on the main stream: order the side stream behind them
        HIPCHK(c, hipEventRecord(c->ev_order, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream_b, c->ev_order, 0));
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
    if (c->scheme == LBM_SCHEME_SLIDE) {
        // the boundary layer the sweep starts with: real update + message; the other one: message only
        hipStream_t bs = c->stream_b ? c->stream_b : c->stream;
        rc = ensure_field_buffers(c);
        if (rc) return rc;
        const int D = c->g.D, want = c->g.fnz > 0;
        const int first = slide_up(c) ? D : 1, last = slide_up(c) ? 1 : D;
        const int skip = c->slide_pristine ? 2 : 0;
        rc = launch_stream_collide<3>(c, slide_in(c), slide_out(c), c->p, first, 1, 0, d_msg_bottom, d_msg_top, want | skip, bs);
        if (!rc && D > 1)
            rc = launch_stream_collide<3>(c, slide_in(c), nullptr, c->p, last, 1, 0, d_msg_bottom, d_msg_top, skip, bs);
        if (rc) return rc;
        if (bs != c->stream) {  // see slide_layers(): the interior's later chunks overwrite what these launches read
            HIPCHK(c, hipEventRecord(c->ev_order, bs));
            c->slide_wait_b = true;
        }
        c->split_phase = 1;
        return LBM_OK;
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
    if (c->scheme == LBM_SCHEME_SLIDE && c->split_phase == 1)
        return fail(c, LBM_ERR_STATE, "lbm_halo_unpack_both: sliding scheme: the ghost layers of the new lattice position are free only after lbm_step_interior_layers");
    const long long per = c->g.mlayer > c->g.H + 2 ? c->g.mlayer : c->g.H + 2;
    float* dst = halo_lattice(c);
    hipStream_t stream = c->stream;
    if (c->pair_phase) {
        // a pair of steps in flight (lbm_pair_*): the messages of its first step go into the ghost layers of the
        // intermediate buffer, those of its second step into the new state's; on the boundary stream, so that they
        // do not queue behind the fused kernel
        if (c->pair_phase == 1) return fail(c, LBM_ERR_STATE, "lbm_halo_unpack_both: call lbm_pair_interior first");
        dst = c->pair_phase == 2 ? c->ft : c->f[1 - c->cur];
        if (c->stream_b) stream = c->stream_b;
    }
    hipLaunchKernelGGL(k_halo_unpack2, dim3(blocks_for(per, 256), 12), dim3(256), 0, stream, dst, d_from_below, d_from_above, c->g,
                       (c->pair_phase && c->side_prio) ? 1 : 0);
    HIPCHK(c, hipGetLastError());
    return LBM_OK;
}

// ---- a PAIR of steps on a z-slab: the fused kernel for the interior, the boundary layers and the rim stepped twice
int lbm_pair_available(lbm_ctx* c, int* available) {
    NEED(c);
    if (!available) return fail(c, LBM_ERR_INVALID, "lbm_pair_available: null argument");
    *available = 0;
    if (!fused_applies_slab(c) || c->split_phase != 0 || c->pair_phase != 0) return LBM_OK;
    const bool was_dirty = c->mask_dirty;
    int rc = refresh_mask_state(c);
    if (!rc) rc = fused_prepare(c);
    if (rc) return rc;
    if (was_dirty && c->stream_b) {  // set-up kernels ran on the main stream: order the boundary stream behind them
        HIPCHK(c, hipEventRecord(c->ev_order, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream_b, c->ev_order, 0));
    }
    *available = c->fused_usable ? 1 : 0;
    return LBM_OK;
}

int lbm_pair_boundary_first(lbm_ctx* c, float* d_msg_bottom, float* d_msg_top) {
    NEED(c);
    if (c->split_phase != 0 || c->pair_phase != 0) return fail(c, LBM_ERR_STATE, "lbm_pair_boundary_first: a split step or a pair is in flight");
    if (!fused_applies_slab(c)) return fail(c, LBM_ERR_STATE, "lbm_pair_boundary_first: pairs need an A/B slab context with lbm_opts.two_step on");
    int avail = 0;
    int rc = lbm_pair_available(c, &avail);
    if (rc) return rc;
    if (!avail) return fail(c, LBM_ERR_STATE, "lbm_pair_boundary_first: the fused path does not apply to this slab (ask lbm_pair_available)");
    if (!c->ev_pair_b1) {
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_pair_b1, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_pair_r1, hipEventDisableTiming));
    }
    hipStream_t bs = c->stream_b ? c->stream_b : c->stream;
    // first step of the two boundary layers: t -> t+1 into the intermediate buffer, messages written as it goes
    rc = launch_stream_collide<2>(c, c->f[c->cur], c->ft, c->p, 1, 2, c->g.D, d_msg_bottom, d_msg_top, 0, bs);
    if (rc) return rc;
    if (bs != c->stream) HIPCHK(c, hipEventRecord(c->ev_pair_b1, bs));
    c->pair_phase = 1;
    return LBM_OK;
}

int lbm_pair_interior(lbm_ctx* c) {
    NEED(c);
    if (c->pair_phase != 1) return fail(c, LBM_ERR_STATE, "lbm_pair_interior: call lbm_pair_boundary_first first");
    hipStream_t bs = c->stream_b ? c->stream_b : c->stream;
    const float* in = c->f[c->cur];
    float* out = c->f[1 - c->cur];
    // The rim's two steps touch no cell the fused kernel touches: they run in front of and behind it on the context's
    // stream.  LBM_PAIR_RIM_STREAM=1 sends them to a third, library-owned stream beside the fused kernel instead
    // (measured at 1024 x 512 x 64: the fused kernel slows by what the rim takes, profiles/r03_tuning.md section 1).
    hipStream_t rs = c->stream;
    if (c->pair_rim_stream && bs != c->stream) {
        if (!c->stream_rim) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->stream_rim, hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_f2_main, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_f2_rim, hipEventDisableTiming));
        }
        rs = c->stream_rim;
        HIPCHK(c, hipEventRecord(c->ev_f2_main, c->stream));          // the previous update, its unpack included
        HIPCHK(c, hipStreamWaitEvent(rs, c->ev_f2_main, 0));
    }
    // the rim's first step before the fused kernel: the boundary stream waits for it alone (the second step of the
    // boundary layers pulls from the layers next to them)
    int rc = launch_stream_collide<4>(c, in, c->ft, c->p, 0, 1, 0, nullptr, nullptr, 0, rs, c->rim_list[0], c->rim_n[0]);
    if (rc) return rc;
    if (bs != c->stream) HIPCHK(c, hipEventRecord(c->ev_pair_r1, rs));
    // the rim's second step pulls from the boundary layers' first step.  With a rim of more than a few per cent it rides
    // in the tail of the fused launch (F2Rim), which then has to start behind that step of the boundary layers -- a short
    // kernel that runs beside the rim's first step
    const bool tail = fused_tail_rim_slab(c);
    if (tail && bs != c->stream) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_pair_b1, 0));
    c->f2_tail_now = tail ? 2 : 0;
    rc = launch_fused_timed(c, in, out);
    c->f2_tail_now = 0;
    if (rc) return rc;
    if (!tail) {
        if (bs != c->stream) HIPCHK(c, hipStreamWaitEvent(rs, c->ev_pair_b1, 0));
        rc = launch_stream_collide<4>(c, c->ft, out, c->p, 0, 1, 0, nullptr, nullptr, 0, rs, c->rim_list[1], c->rim_n[1]);
        if (rc) return rc;
    }
    if (rs != c->stream) {   // whatever follows on the context's stream sees the rim's results
        HIPCHK(c, hipEventRecord(c->ev_f2_rim, rs));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_f2_rim, 0));
    }
    c->pair_phase = 2;
    return LBM_OK;
}

int lbm_pair_boundary_second(lbm_ctx* c, float* d_msg_bottom, float* d_msg_top) {
    NEED(c);
    if (c->pair_phase != 2) return fail(c, LBM_ERR_STATE, "lbm_pair_boundary_second: call lbm_pair_interior (and unpack the first step's messages) first");
    hipStream_t bs = c->stream_b ? c->stream_b : c->stream;
    if (bs != c->stream) HIPCHK(c, hipStreamWaitEvent(bs, c->ev_pair_r1, 0));
    // second step of the two boundary layers: t+1 -> t+2, from the intermediate buffer (ghost layers just unpacked)
    // (beside the fused kernel: in workgroups of one wave, which find room on the CUs that kernel fills)
    c->wave_blocks_now = c->pair_wave_blocks;
    int rc = launch_stream_collide<2>(c, c->ft, c->f[1 - c->cur], c->p, 1, 2, c->g.D, d_msg_bottom, d_msg_top, 0, bs);
    c->wave_blocks_now = false;
    if (rc) return rc;
    c->pair_phase = 3;
    return LBM_OK;
}

int lbm_pair_finish(lbm_ctx* c) {
    NEED(c);
    if (c->pair_phase != 3) return fail(c, LBM_ERR_STATE, "lbm_pair_finish: call lbm_pair_boundary_second first");
    c->cur = 1 - c->cur;
    c->p_last = c->p;
    c->steps += 2;
    c->fields = FIELDS_STALE;
    c->pair_last = true;   // the previous buffer is two steps old: no rho/u on demand until a single step has run
    c->ended_on_pair = false;
    c->prev_touched = false;
    c->curl_valid = false;
    c->pair_phase = 0;
    return LBM_OK;
}

int lbm_step_interior_layers(lbm_ctx* c) {
    NEED(c);
    if (c->split_phase != 1) return fail(c, LBM_ERR_STATE, "lbm_step_interior_layers: call lbm_step_boundary_layers first");
    int rc;
    if (c->scheme == LBM_SCHEME_SLIDE) {
        const int D = c->g.D, want = c->g.fnz > 0;
        rc = slide_layers(c, 2, D - 2, want);
        if (!rc && c->slide_wait_b) {  // fewer than two interior chunks: nothing waited for the boundary launches yet
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_order, 0));
            c->slide_wait_b = false;
        }
        // the boundary layer the sweep ends with (its message left in lbm_step_boundary_layers)
        if (!rc && D > 1)
            rc = launch_stream_collide<3>(c, slide_in(c), slide_out(c), c->p, slide_up(c) ? 1 : D, 1, 0, nullptr, nullptr,
                                          want | (c->slide_pristine ? 2 : 0));
    } else {
        rc = c->scheme == LBM_SCHEME_AA ? launch_aa_layers(c, c->g.fnz > 0, 2, c->g.D - 2)
                                        : launch_ab_layers(c, 2, c->g.D - 2);
    }
    if (rc) return rc;
    c->split_phase = 2;
    return LBM_OK;
}

int lbm_step_finish(lbm_ctx* c) {
    NEED(c);
    if (c->split_phase != 2) return fail(c, LBM_ERR_STATE, "lbm_step_finish: call lbm_step_interior_layers first");
    if (c->scheme == LBM_SCHEME_AA) aa_flip(c);
    else if (c->scheme == LBM_SCHEME_SLIDE) slide_flip(c);
    else c->cur = 1 - c->cur;
    c->p_last = c->p;
    c->steps += 1;
    c->fields = c->scheme == LBM_SCHEME_AB ? FIELDS_STALE : FIELDS_VALID;
    c->prev_touched = false;
    c->pair_last = false;
    c->ended_on_pair = false;
    c->curl_valid = false;
    c->split_phase = 0;
    return LBM_OK;
}

// ------------------------------------------------------------------------------------------
// stream-ordered flags between ranks (cfd_amd/slab.py: IpcCopyP2P)
// ------------------------------------------------------------------------------------------
// A halo transport made of peer copies has no kernels of its own competing with k_fused2 for the CUs (RCCL's send / recv
// kernel does: profiles/r03_tuning.md section 8), but the two ranks of a link then need another way to tell each other
// "the data has landed" and "the buffer is free again" in stream order.  The flags a rank waits on live in fine-grained
// memory of its own GPU that its neighbours map through an IPC handle: the writer's store goes out when its stream gets
// there, the reader's stream holds a one-lane kernel that polls until the value is reached -- or a time limit passes,
// so the grid always drains.
__global__ void k_flag_post(unsigned* flag, unsigned value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_flag_await(const unsigned* flag, unsigned value, unsigned long long ticks, unsigned* timed_out) {
    // (an earlier wait of this rank gave up: the peer is gone, do not wait again for every exchange still queued)
    if (__hip_atomic_load(timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const unsigned v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int)(v - value) >= 0) return;
        if (wall_clock64() - t0 > ticks) {
            __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        __builtin_amdgcn_s_sleep(64);
    }
}

// several posts and several waits in one launch: the posts first (see IpcCopyP2P: every rank posts before it waits)
struct FlagOps {
    unsigned* post[LBM_FLAG_OPS_MAX];
    const unsigned* wait[LBM_FLAG_OPS_MAX];
    unsigned post_value[LBM_FLAG_OPS_MAX], wait_value[LBM_FLAG_OPS_MAX];
    int nposts, nwaits;
};
__global__ void k_flag_sync(FlagOps ops, unsigned long long ticks, unsigned* timed_out) {
    for (int i = 0; i < ops.nposts; i++) __hip_atomic_store(ops.post[i], ops.post_value[i], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (ops.nwaits == 0) return;
    if (__hip_atomic_load(timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < ops.nwaits; i++) {
        for (;;) {
            const unsigned v = __hip_atomic_load(ops.wait[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int)(v - ops.wait_value[i]) >= 0) break;
            if (wall_clock64() - t0 > ticks) {
                __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
            __builtin_amdgcn_s_sleep(64);
        }
    }
}

int lbm_ipc_alloc(size_t bytes, int fine_grained, void** d_ptr, void* ipc_handle) {
    if (!bytes || !d_ptr) return fail(nullptr, LBM_ERR_INVALID, "lbm_ipc_alloc: null argument");
    *d_ptr = nullptr;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "include/lbm_abi.h documents a 64-byte handle");
    // fine-grained device memory (the flags): a word a peer stores into is seen by a kernel that is already polling it.
    // The message buffers are ordinary device memory: their reader starts after the flag said the data has landed.
    void* p = nullptr;
    hipError_t e = fine_grained ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) : hipMalloc(&p, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, LBM_ERR_HIP, "lbm_ipc_alloc: %s", hipGetErrorString(e)); }
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && ipc_handle) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(ipc_handle), p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        return fail(nullptr, LBM_ERR_HIP, "lbm_ipc_alloc: %s", hipGetErrorString(e));
    }
    *d_ptr = p;
    return LBM_OK;
}

int lbm_ipc_free(void* d_ptr) {
    if (!d_ptr) return LBM_OK;
    const hipError_t e = hipFree(d_ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, LBM_ERR_HIP, "lbm_ipc_free: %s", hipGetErrorString(e)); }
    return LBM_OK;
}

int lbm_ipc_open(const void* ipc_handle, void** d_ptr) {
    if (!ipc_handle || !d_ptr) return fail(nullptr, LBM_ERR_INVALID, "lbm_ipc_open: null argument");
    *d_ptr = nullptr;
    hipIpcMemHandle_t h;
    memcpy(&h, ipc_handle, sizeof h);
    const hipError_t e = hipIpcOpenMemHandle(d_ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError(); *d_ptr = nullptr; return fail(nullptr, LBM_ERR_HIP, "lbm_ipc_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e)); }
    return LBM_OK;
}

int lbm_ipc_close(void* d_ptr) {
    if (!d_ptr) return LBM_OK;
    const hipError_t e = hipIpcCloseMemHandle(d_ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, LBM_ERR_HIP, "lbm_ipc_close: %s", hipGetErrorString(e)); }
    return LBM_OK;
}

int lbm_ipc_copy(void* stream, void* d_dst, const void* d_src, size_t bytes) {
    if (!d_dst || !d_src) return fail(nullptr, LBM_ERR_INVALID, "lbm_ipc_copy: null argument");
    if (!bytes) return LBM_OK;
    const hipError_t e = hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, LBM_ERR_HIP, "lbm_ipc_copy: %s", hipGetErrorString(e)); }
    return LBM_OK;
}

int lbm_flag_sync(void* stream, uint32_t* const* d_post, const uint32_t* post_values, int nposts,
                  const uint32_t* const* d_wait, const uint32_t* wait_values, int nwaits, uint32_t timeout_ms, uint32_t* d_timed_out) {
    if (nposts < 0 || nposts > LBM_FLAG_OPS_MAX || nwaits < 0 || nwaits > LBM_FLAG_OPS_MAX)
        return fail(nullptr, LBM_ERR_INVALID, "lbm_flag_sync: %d posts, %d waits (0 .. %d each)", nposts, nwaits, LBM_FLAG_OPS_MAX);
    if ((nposts && (!d_post || !post_values)) || (nwaits && (!d_wait || !wait_values || !d_timed_out)))
        return fail(nullptr, LBM_ERR_INVALID, "lbm_flag_sync: null argument");
    if (nwaits && (timeout_ms == 0 || timeout_ms > 600000u)) return fail(nullptr, LBM_ERR_INVALID, "lbm_flag_sync: timeout_ms %u (1 .. 600000)", timeout_ms);
    if (nposts + nwaits == 0) return LBM_OK;
    FlagOps ops;
    memset(&ops, 0, sizeof ops);
    ops.nposts = nposts; ops.nwaits = nwaits;
    for (int i = 0; i < nposts; i++) {
        if (!d_post[i]) return fail(nullptr, LBM_ERR_INVALID, "lbm_flag_sync: null flag");
        ops.post[i] = d_post[i]; ops.post_value[i] = post_values[i];
    }
    for (int i = 0; i < nwaits; i++) {
        if (!d_wait[i]) return fail(nullptr, LBM_ERR_INVALID, "lbm_flag_sync: null flag");
        ops.wait[i] = d_wait[i]; ops.wait_value[i] = wait_values[i];
    }
    hipLaunchKernelGGL(k_flag_sync, dim3(1), dim3(1), 0, (hipStream_t)stream, ops, (unsigned long long)timeout_ms * 100000ull, d_timed_out);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LBM_OK : fail(nullptr, LBM_ERR_HIP, "lbm_flag_sync: %s", hipGetErrorString(e));
}

int lbm_flag_peek(const uint32_t* d_flag, uint32_t* value) {
    if (!d_flag || !value) return fail(nullptr, LBM_ERR_INVALID, "lbm_flag_peek: null argument");
    const hipError_t e = hipMemcpy(value, d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, LBM_ERR_HIP, "lbm_flag_peek: %s", hipGetErrorString(e)); }
    return LBM_OK;
}

int lbm_flag_post(void* stream, uint32_t* d_flag, uint32_t value) {
    if (!d_flag) return fail(nullptr, LBM_ERR_INVALID, "lbm_flag_post: null flag");
    hipLaunchKernelGGL(k_flag_post, dim3(1), dim3(1), 0, (hipStream_t)stream, d_flag, value);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LBM_OK : fail(nullptr, LBM_ERR_HIP, "lbm_flag_post: %s", hipGetErrorString(e));
}

int lbm_flag_await(void* stream, const uint32_t* d_flag, uint32_t value, uint32_t timeout_ms, uint32_t* d_timed_out) {
    if (!d_flag || !d_timed_out) return fail(nullptr, LBM_ERR_INVALID, "lbm_flag_await: null argument");
    if (timeout_ms == 0 || timeout_ms > 600000u) return fail(nullptr, LBM_ERR_INVALID, "lbm_flag_await: timeout_ms %u (1 .. 600000)", timeout_ms);
    hipLaunchKernelGGL(k_flag_await, dim3(1), dim3(1), 0, (hipStream_t)stream, d_flag, value,
                       (unsigned long long)timeout_ms * 100000ull /* 100 MHz */, d_timed_out);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LBM_OK : fail(nullptr, LBM_ERR_HIP, "lbm_flag_await: %s", hipGetErrorString(e));
}

}  // extern "C"
