/*
 * lbm_abi.h -- C ABI of the MI355X-native D3Q19 lattice-Boltzmann hot path (liblbm_hip.so).
 *
 * The reference (1neskk/CFD) has no FFI: its solver is a C++ class compiled into the engine
 * (Engine/src/Physics/lbm_solver.cuh:17-80).  Each entry point below replaces one member of that
 * class (cited as file:line into the reference); the header-only C++ class
 * cfd_amd/csrc/lbm_solver.hpp puts the reference's own names back on top of this ABI so the
 * engine's `cfd` layer (Engine/src/main.cpp:17-28,36) keeps compiling against it headless.
 *
 * Conventions
 *   - plain C: opaque context pointer, ints, floats, raw host pointers and byte/element counts.
 *   - every call returns 0 on success or a negative lbm_status; lbm_last_error() gives the text.
 *   - a context is bound to one HIP device and one stream; it is not thread-safe.
 *   - host arrays use the reference's wire formats, x fastest: idx = z*W*H + y*W + x
 *     (lbm_solver.cu:253): density float[n], velocity float[n][4] = (ux,uy,uz,0)
 *     (cuda_math.cuh:37-55, lbm_solver.cu:337), solid uint8[n] with nonzero = solid (255 written,
 *     lbm_solver.cu:511), curl float[n].
 *   - lbm_step() is asynchronous on the context's stream like the reference's launches
 *     (lbm_solver.cu:482-493); read-backs and lbm_sync() synchronise.
 *   - there is no CPU fallback: every compute entry point fails with LBM_ERR_NO_DEVICE when no
 *     HIP device is usable.
 */
#ifndef LBM_ABI_H
#define LBM_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBM_ABI_VERSION 3
#define LBM_Q 19

typedef struct lbm_ctx lbm_ctx;

typedef enum lbm_status {
    LBM_OK = 0,
    LBM_ERR_INVALID = -1,    /* bad argument (null pointer, non-positive size, count mismatch) */
    LBM_ERR_NO_DEVICE = -2,  /* no usable HIP device */
    LBM_ERR_HIP = -3,        /* a HIP runtime call failed; see lbm_last_error */
    LBM_ERR_IO = -4,         /* mesh file could not be read */
    LBM_ERR_STATE = -5       /* call not valid in the current state (e.g. slab call on a non-slab ctx) */
} lbm_status;

/* Boundary presets.  WIND_TUNNEL_REF is the reference's hard-wired chain
 * (lbm_solver.cu:283-318: inlet x<0, outlet x>=W, floor bounce-back, ceiling reflect_y,
 * side reflect_z, solid bounce-back, in that order, quirks included).  CAVITY is a closed box
 * with a lid at y=H-1 moving in +x with `inlet_velocity`; it is NOT in the reference and exists
 * for the pure-roofline configuration (BASELINE.json configs[1]). */
enum { LBM_PRESET_WIND_TUNNEL_REF = 0, LBM_PRESET_CAVITY = 1 };

/* Streaming schemes.  AB = pull, two population buffers (lbm_solver.cu:482-490).  AA = one
 * buffer updated in place (even step: own cell read / swapped own cell write; odd step:
 * neighbour read / neighbour write): same results, half the population memory; rho/u are written
 * by the last step of each lbm_step() call; a cell that a later mask change turns from solid back
 * to fluid restarts from equilibrium (there is no second buffer to keep its frozen values).
 * Both schemes run as z-slabs.
 * SLIDE = the A/B kernel on ONE buffer with chunk + 2 spare layers: a step reads the lattice at one
 * layer offset and writes it at the other, a chunk of layers per launch, in the direction of the
 * shift, so a launch never overwrites what it or a later launch still reads.  Same results, the
 * A/B kernel's aligned access pattern, 1/2 + (chunk + 2)/(2 D) of the memory; a solid cell's frozen
 * populations move with the lattice (or, while every solid cell still holds what lbm_init() wrote, are
 * not stored at all), so every result equals A/B's, solid cells included; rho/u are
 * written by the last step of each lbm_step() call as in AA.  It runs as z-slabs with the same calls
 * and the A/B message format; lbm_halo_unpack / lbm_halo_unpack_both must then come AFTER
 * lbm_step_interior_layers (the ghost layers of the new lattice position are input until then). */
enum { LBM_SCHEME_AB = 0, LBM_SCHEME_AA = 1, LBM_SCHEME_SLIDE = 2 };

/* Population storage in HBM (fp32 SoA rows padded to 128-byte lines either way):
 *   PLANES [k][z][y][x]  19 separate planes, skewed against each other;
 *   ROWS   [z][y][k][x]  the 19 runs of one (y,z) row interleaved into one contiguous chunk --
 *                        measured 6-7 % faster under the pull pattern (profiles/r01_tuning.md). */
enum { LBM_LAYOUT_DEFAULT = 0, LBM_LAYOUT_PLANES = 1, LBM_LAYOUT_ROWS = 2 };

typedef struct lbm_opts {
    int32_t struct_size;   /* = sizeof(lbm_opts); lets the ABI grow */
    int32_t device;        /* HIP device ordinal; -1 = current device */
    int32_t preset;        /* LBM_PRESET_* */
    int32_t scheme;        /* LBM_SCHEME_* */
    /* z-slab decomposition (multi-GPU): this context owns global layers [z0, z0+d) of a domain
     * that is depth_global deep.  depth_global = 0 means "not decomposed" (z0 = 0, depth = d). */
    int32_t z0;
    int32_t depth_global;
    void*   stream;        /* hipStream_t to run on; NULL = the context creates its own */
    int32_t plane_skew_bytes; /* LBM_LAYOUT_PLANES: extra bytes between population planes; -1 = library default */
    int32_t layout;        /* LBM_LAYOUT_*; 0 = library default (row-interleaved) */
    /* optional second hipStream_t (ideally high priority) for lbm_step_boundary_layers, so the
     * boundary layers and the halo exchange run beside the interior update instead of in front of
     * it; the CALLER orders the two streams (see cfd_amd/slab.py).  NULL = use `stream`. */
    void*   boundary_stream;
    int32_t slide_chunk;   /* LBM_SCHEME_SLIDE: layers per launch (spare layers = chunk + 2); 0 = library default (32) */
    /* LBM_SCHEME_AB, undecomposed, row-interleaved layout: lbm_step(ctx, n >= 2) advances TWO time steps per pass
     * over HBM wherever a block of tiles keeps off the domain's walls, obstacles included (k_fused2.h: 76 B x
     * (1 + halo) read + 76 B written per two updates instead of 304 B) and takes the rim -- wall rows and layers --
     * through two ordinary steps via a third population buffer (allocated on first use: +50 % population memory;
     * with obstacles also one 32-bit word per cell).  Results are bit-identical
     * to single steps.  The last step of a call is a single step when n is odd or when the caller read rho/u after its
     * previous call (that step then writes them); an even n without that runs in pairs to the end, and rho/u of the
     * last step stay available on demand (at the price of one more step).
     * LBM_TWO_STEP_AUTO = on where it applies. */
    int32_t two_step;      /* LBM_TWO_STEP_* */
    int32_t reserved[2];
} lbm_opts;
enum { LBM_TWO_STEP_AUTO = 0, LBM_TWO_STEP_OFF = 1 };

/* Per-context counters for benches (device time is measured with HIP events on the
 * context's stream around the lbm_step calls issued between timer_begin and timer_end). */
typedef struct lbm_stats {
    uint64_t steps;             /* steps issued since create */
    uint64_t cells;             /* W*H*D owned by this context */
    uint64_t bytes_per_cell;    /* algorithmic bytes per cell update of the step kernel (152) */
    uint64_t population_bytes;  /* device bytes held by the population buffers */
    double   last_timed_ms;     /* device ms between the last timer_begin/timer_end pair */
    uint64_t last_timed_steps;  /* steps issued between them */
    uint64_t last_timed_launches; /* one-step kernel launches issued between them (large domains step in several
                                   * launches of about 1 GB of traffic each, slabs in boundary + interior launches);
                                   * the launches of the two-steps-per-pass path are counted below instead */
    /* two steps per pass (lbm_opts.two_step): what one k_fused2 launch covers, and its share of the timed region */
    uint64_t fused_cells;         /* cells one k_fused2 launch advances by two steps (0: the fused path is not in use) */
    uint64_t rim_cells;           /* cells of the rim tiles, which take two one-step launches per pair instead */
    uint64_t last_timed_fused_launches; /* k_fused2 launches between timer_begin and timer_end */
    double   last_timed_fused_ms; /* their summed device time (HIP events around each launch) */
    /* the form of k_fused2's first-step loads for blocks without obstacles in reach: timed on the first launch after a
     * plan (LBM_FUSED_WIDE unset), neither form wins everywhere */
    int32_t  fused_wide_loads;    /* 1 wide (16-byte loads of whole rows), 0 narrow, -1 not decided yet */
    float    fused_tuned_ms[2];   /* what one launch took when they were timed: narrow, wide (0: never timed) */
    /* cells that the SAME launch advances by ONE step: the rim's first step, when it rides in the tail of the fused
     * launch (undecomposed contexts whose rim is more than a few per cent of the cells; 0 otherwise) */
    uint64_t fused_tail_cells;
} lbm_stats;

/* lbm_solver::lbm_solver(int,int,int)  -- lbm_solver.cu:404-453.  opts may be NULL. */
int lbm_create(lbm_ctx** out, int width, int height, int depth, const lbm_opts* opts);
/* lbm_solver::~lbm_solver  -- lbm_solver.cu:455-469 */
int lbm_destroy(lbm_ctx* ctx);

/* lbm_solver::init / reset  -- lbm_solver.cu:471-475, 496 (k_init :37-64): f_k = w_k, rho = 1,
 * u = 0; the solid mask is kept. */
int lbm_init(lbm_ctx* ctx);

/* settings{inlet_velocity, omega_shear, omega_bulk} -- lbm_solver.cuh:9-15, read at every launch
 * (lbm_solver.cu:485-486).  `tau` is dead in the reference and has no counterpart. */
int lbm_set_params(lbm_ctx* ctx, float inlet_velocity, float omega_shear, float omega_bulk);
int lbm_get_params(lbm_ctx* ctx, float* inlet_velocity, float* omega_shear, float* omega_bulk);

/* lbm_solver::add_solid(Rect) -- lbm_solver.cu:498-519: half-open box, sets, never clears. */
int lbm_add_solid_rect(lbm_ctx* ctx, int x, int y, int z, int w, int h, int d);
/* d_solid.copy_from_host -- cuda_buffer.cuh:72-77 as used at lbm_solver.cu:555: replaces the
 * whole mask (nonzero = solid).  For a slab context `n` is the slab's own W*H*depth cells. */
int lbm_set_solid_u8(lbm_ctx* ctx, const uint8_t* mask, size_t n);
/* cuda_buffer<unsigned char>::copy_from_device on d_solid -- cuda_buffer.cuh:86-91: the whole mask from a dense
 * uint8 array in DEVICE memory (nonzero = solid); synchronises like the reference's cudaMemcpy. */
int lbm_set_solid_u8_device(lbm_ctx* ctx, const uint8_t* d_mask, size_t n);
/* cuda_buffer<T>::copy_to_device -- cuda_buffer.cuh:93-98: a blocking device-to-device copy on the context's
 * device and stream, for hosts that do not link the HIP runtime themselves (the source is a pointer obtained
 * from lbm_device_fields). */
int lbm_device_copy(lbm_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
/* cuda_buffer<T>::copy_to_host with fewer elements than the field holds -- cuda_buffer.cuh:79-84: a blocking
 * device-to-host copy of `bytes` from a pointer obtained from lbm_device_fields. */
int lbm_copy_to_host(lbm_ctx* ctx, void* dst, const void* d_src, size_t bytes);
/* cuda_buffer<T>::copy_from_host / copy_from_device / memset on d_rho, d_u (vec4) and d_curl -- cuda_buffer.cuh:66-77,
 * 86-91.  These arrays are outputs of step() in the reference (lbm_solver.cu:336-337, :655): what a caller writes is
 * what readers see until the next step rewrites the array, and it never reaches the populations.  Same here: the
 * first `count` elements of the field (of the current field window for density / velocity) are overwritten, from
 * host memory or (src_on_device != 0) from device memory; the next lbm_step() supersedes them. */
enum { LBM_FIELD_DENSITY = 0, LBM_FIELD_VELOCITY = 1, LBM_FIELD_CURL = 2 };
int lbm_write_field(lbm_ctx* ctx, int field, const void* src, size_t count, int src_on_device);
/* lbm_solver::load_mesh -- lbm_solver.cu:521-556 + MeshVoxelizer (mesh_voxelizer.cpp:20-246):
 * OBJ -> rotate_y(-90) -> 0.8 auto-scale -> +X ray-cast voxelisation -> y=0 plane solid ->
 * whole mask replaced.  A missing file returns LBM_ERR_IO after installing the floor-only mask,
 * which is what the reference's "log and carry on" leaves behind (:522-524). */
int lbm_load_mesh(lbm_ctx* ctx, const char* obj_path);

/* lbm_solver::step -- lbm_solver.cu:477-494: `nsteps` fused stream-collide updates with the A/B
 * swap.  rho/u/curl are not rewritten every step; they are materialised, bit-identical to what
 * the reference's per-step writes would hold, when read (lbm_read_*, lbm_device_fields). */
int lbm_step(lbm_ctx* ctx, int nsteps);
/* The reference's frame loop kept on the device -- lbm_solver::step as the engine calls it once per frame
 * (Engine/src/main.cpp:36): every step rewrites rho and u (lbm_solver.cu:336-337) and runs the curl pass (:488).
 * `nframes` such steps back to back on the context's stream, no host round trip between them; afterwards rho, u
 * (and curl if want_curl != 0) are those of the last frame, bit for bit what nframes x { lbm_step(ctx, 1);
 * lbm_device_fields(...) } leaves.  172 B per cell and frame, 192 B with the curl. */
int lbm_step_frames(lbm_ctx* ctx, int nframes, int want_curl);
/* lbm_solver::compute_curl -- lbm_solver.cu:612-661 */
int lbm_compute_curl(lbm_ctx* ctx);
/* cudaDeviceSynchronize of the debug build / renderer -- lbm_solver.cu:491-493 */
int lbm_sync(lbm_ctx* ctx);

/* get_density().copy_to_host etc. -- lbm_solver.cuh:40-46 + cuda_buffer.cuh:79-84.
 * `n` is the element count of the host array and must equal W*H*depth of this context. */
int lbm_read_density(lbm_ctx* ctx, float* rho, size_t n);
int lbm_read_velocity(lbm_ctx* ctx, float* u_xyzw, size_t n);
int lbm_read_solid_u8(lbm_ctx* ctx, uint8_t* mask, size_t n);
int lbm_read_curl(lbm_ctx* ctx, float* curl, size_t n);
/* Derived fields for the consumers downstream of the path (the reference's renderer samples u as an
 * RGBA32F volume and colours by |u|: renderer.cpp:1382-1389, Shaders/raymarch.frag:93-128,
 * Shaders/streamline.comp:36-62).  Both are computed on the device from the fields of the current
 * field window (nz layers):
 *   lbm_read_velocity_strided: u at every (i*sx, j*sy, k*sz), n = ceil(W/sx)*ceil(H/sy)*ceil(nz/sz)
 *   samples of 4 floats, x fastest -- a down-sampled volume without moving the full field to the host;
 *   lbm_read_speed: |u.xyz| per cell (cuda_math::length, cuda_math.cuh:142-145), n = W*H*nz. */
int lbm_read_velocity_strided(lbm_ctx* ctx, int sx, int sy, int sz, float* u_xyzw, size_t n);
int lbm_read_speed(lbm_ctx* ctx, float* speed, size_t n);
/* lbm_read_velocity_strided for callers that cannot allocate: a renderer's per-frame loop (hipMalloc / hipFree
 * synchronise the device), serving loops and harnesses that forbid allocation.  It still synchronises the
 * context's stream and copies to (pageable) host memory, so it is NOT a call for code under HIP graph capture.  No
 * counterpart in the reference.  Same arguments and result; the sampling kernel's staging area is a fixed
 * array inside the library's code object (LBM_NOALLOC_MAX_SAMPLES samples of 4 floats = 4 MiB, held by
 * every GPU the library runs on until the process ends) instead of a hipMalloc per call, and the call
 * issues no hipMalloc / hipFree:
 *   - n > LBM_NOALLOC_MAX_SAMPLES is rejected with LBM_ERR_INVALID (use larger strides);
 *   - rho/u are allocated on first use, which this call may not do: it returns LBM_ERR_STATE until some
 *     field has been read once (lbm_read_velocity, lbm_device_fields, ...) after lbm_create or
 *     lbm_set_field_window;
 *   - all contexts on one GPU share that GPU's array: calls that overlap on one GPU (two host threads,
 *     two contexts) race.  The kernel writes every sample it returns, so nothing is cleared between calls. */
#define LBM_NOALLOC_MAX_SAMPLES (1 << 18)
int lbm_read_velocity_strided_noalloc(lbm_ctx* ctx, int sx, int sy, int sz, float* u_xyzw, size_t n);
/* Current populations in the reference layout f[k*n + idx] (d_f, lbm_solver.cuh:62). */
int lbm_read_populations(lbm_ctx* ctx, float* f, size_t n_times_19);
int lbm_write_populations(lbm_ctx* ctx, const float* f, size_t n_times_19);
/* Which branch of lbm_solver.cu:283-318 feeds each direction of each cell: cls[idx*19 + k] in
 * {0 pull, 1 inlet, 2 outlet, 3 floor-BB, 4 ceiling reflect_y, 5 side reflect_z, 6 solid-BB,
 *  7 wall-BB (cavity), 8 lid-BB (cavity)}; 0 for every k of a solid cell. */
int lbm_read_classification(lbm_ctx* ctx, uint8_t* cls, size_t n_times_19);

/* Device pointers of the fields in the reference wire formats, brought up to date and
 * synchronised, for consumers that stay on the GPU (the reference hands d_u/d_solid to its
 * renderer: get_velocity()/get_solid_mask(), lbm_solver.cuh:40-46; renderer.cpp:1382-1389).
 * Any output pointer may be NULL, and what is not asked for is not computed (curl: 20 B/cell).
 * d_solid is a dense uint8 copy (0/255) of the packed mask. */
int lbm_device_fields(lbm_ctx* ctx, float** d_rho, float** d_u_xyzw, float** d_curl, uint8_t** d_solid);

/* Domains sized to HBM (BASELINE.json configs[4]: 266 GB of populations on a 288 GB GPU) have
 * no room for whole-domain rho/u (20 B/cell, where the reference allocates them up front,
 * lbm_solver.cu:410-411).  The field window restricts rho/u to the owned layers
 * [z_begin, z_begin+nz): lbm_read_density / lbm_read_velocity / lbm_device_fields then take and
 * return W*H*nz elements, so a caller can walk the domain window by window (A/B scheme: any
 * window of the last step is materialised on demand; AA scheme: the window is filled by the next
 * lbm_step).  nz = 0 keeps no fields at all; the default is the whole domain.  lbm_compute_curl
 * needs the whole domain in the window.  No counterpart in the reference. */
int lbm_set_field_window(lbm_ctx* ctx, int z_begin, int nz);

/* Rebinds the context to other caller-owned streams (lbm_opts.stream / boundary_stream) after
 * draining the current ones.  HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by
 * default); two streams that share a queue do not overlap, so a multi-GPU driver times a few steps
 * on candidate streams and keeps the pair on which the boundary layers, the RCCL exchange and the
 * interior update really run side by side (cfd_amd/slab.py: SlabSolver.autotune_streams).
 * No counterpart in the reference (single default stream). */
int lbm_set_streams(lbm_ctx* ctx, void* stream, void* boundary_stream);

int lbm_get_dims(lbm_ctx* ctx, int* w, int* h, int* d);
int lbm_get_stats(lbm_ctx* ctx, lbm_stats* out);
int lbm_timer_begin(lbm_ctx* ctx);
int lbm_timer_end(lbm_ctx* ctx);   /* synchronises the stream and fills last_timed_* */
const char* lbm_last_error(lbm_ctx* ctx); /* ctx may be NULL: error of the last failed create */
int lbm_abi_version(void);

/* ---- host-only entry points: no device is touched --------------------------------------
 * The voxelisation stage of lbm_solver::load_mesh alone (lbm_solver.cu:526-553 +
 * mesh_voxelizer.cpp:146-216): writes the reference's uint8 mask (255 = solid, floor included)
 * for a width x height x depth grid into `mask` (n = W*H*D bytes) and the auto-scale into
 * *scale_out.  The reference runs this stage on the host too. */
int lbm_host_voxelize_obj(const char* obj_path, int width, int height, int depth,
                          uint8_t* mask, size_t n, float* scale_out);
/* One cell of lbm_solver.cu:320-401 evaluated on the host with the very inline functions the
 * kernels use (cfd_amd/csrc/d3q19.h): moments of f19, then the collision.  Diagnostic used to
 * prove, without a GPU, that the folded arithmetic equals the reference's literal loops. */
int lbm_host_collide_cell(const float* f19, float omega_shear, float omega_bulk,
                          float* out19, float* rho_u4);

/* ---- z-slab halo exchange (multi-GPU; the reference is single-GPU) ----
 * A context created with lbm_opts.depth_global > 0 is one z-slab of a taller domain.  After each
 * step it sends the five populations that cross its top face (k = 5,11,14,15,18) from its top
 * owned layer to the upper neighbour's bottom ghost layer, and the five that cross its bottom
 * face (k = 6,12,13,16,17) from its bottom owned layer to the lower neighbour's top ghost layer.
 * One population layer is one contiguous run of floats, so a message is five such runs plus a
 * short tail with the outlet-edge values that the reference's unchecked outlet index reads across
 * a face (quirk Q2, lbm_solver.cu:292-296).  Because that index also wraps from the last layer to
 * the first, the wind-tunnel preset exchanges around a RING: rank r sends its top message to
 * (r+1) mod P and its bottom message to (r-1) mod P; the cavity preset needs only the open chain.
 *   side: 0 = bottom face (towards z-1), 1 = top face (towards z+1).
 * lbm_halo_pack gathers this slab's outgoing message for `side` into caller-owned DEVICE memory
 * of lbm_halo_message_floats() floats; lbm_halo_unpack scatters the message received from the
 * neighbour on `side` into that side's ghost layer.  The transport in between is the caller's
 * (RCCL send/recv via torch.distributed in cfd_amd/slab.py; a device copy in single-GPU tests).
 * Both run on the context's stream.
 * With LBM_SCHEME_AA the same calls apply; what a message carries follows the layout of the moment
 * (after an even step the boundary layer's own slots go to the neighbour's ghost layer, after an odd
 * step the values pushed into the ghost layer go to the neighbour's boundary layer) and its tail is
 * the outlet-column side array of the boundary layer (7*H floats). */
int lbm_halo_message_floats(lbm_ctx* ctx, size_t* nfloats);
int lbm_halo_pack(lbm_ctx* ctx, int side, float* d_message);
int lbm_halo_unpack(lbm_ctx* ctx, int side, const float* d_message);
/* Static ghost layers of the solid mask (exchanged once after the mask changes): this slab's own
 * boundary layer on `side` as uint8[W*H], and the neighbour's layer to install as the ghost. */
int lbm_halo_get_mask_layer(lbm_ctx* ctx, int side, uint8_t* layer, size_t n);
int lbm_halo_set_mask_ghost(lbm_ctx* ctx, int side, const uint8_t* layer, size_t n);
/* Both received messages in one launch (either may be NULL). */
int lbm_halo_unpack_both(lbm_ctx* ctx, const float* d_from_below, const float* d_from_above);
/* One step split for overlap: the two boundary layers first (their results feed the messages),
 * then the interior layers while the messages travel, then the A/B swap.
 * lbm_step(ctx, 1) == boundary + interior + finish.  lbm_step_boundary_layers also writes the
 * outgoing bottom / top messages (what lbm_halo_pack(0/1) would gather afterwards) straight into
 * the given device buffers when they are not NULL, saving the pack passes. */
int lbm_step_boundary_layers(lbm_ctx* ctx, float* d_msg_bottom, float* d_msg_top);
int lbm_step_interior_layers(lbm_ctx* ctx);
int lbm_step_finish(lbm_ctx* ctx);

/* TWO steps per exchange pattern on a slab (lbm_opts.two_step, A/B, row-interleaved layout): the interior advances
 * by two steps in one pass (k_fused2), only the two boundary layers and the rim are stepped twice, and each of
 * their steps is followed by the usual exchange -- the same messages as two single split steps, with the fused
 * kernel running beside both exchanges.  One pair =
 *     lbm_pair_boundary_first(msg_b, msg_t)    boundary layers t -> t+1, writes the outgoing messages   [boundary stream]
 *     lbm_pair_interior()                      rim t -> t+1, fused interior t -> t+2, rim t+1 -> t+2    [stream]
 *     <exchange>, lbm_halo_unpack_both(...)    the neighbours' t+1 layers (the library routes them into the
 *                                              intermediate buffer, on the boundary stream)
 *     lbm_pair_boundary_second(msg_b, msg_t)   boundary layers t+1 -> t+2, messages                     [boundary stream]
 *     <exchange>, lbm_halo_unpack_both(...)    the neighbours' t+2 layers into the new state's ghost layers
 *     lbm_pair_finish()                        after the caller has joined the two streams
 * The library orders its own launches across the two streams; the CALLER orders the exchange behind the boundary
 * launches and joins the streams before lbm_pair_finish (cfd_amd/slab.py: SlabSolver._pair).  Results equal two
 * single steps bit for bit.  After a pair rho/u cannot be recomputed on demand (the previous buffer is two steps
 * old): finish a batch with a single split step, as lbm_step() does for undecomposed contexts.
 * lbm_pair_available reports (after bringing the tile lists up to date) whether the fused path applies to this slab. */
int lbm_pair_available(lbm_ctx* ctx, int* available);
int lbm_pair_boundary_first(lbm_ctx* ctx, float* d_msg_bottom, float* d_msg_top);
int lbm_pair_interior(lbm_ctx* ctx);
int lbm_pair_boundary_second(lbm_ctx* ctx, float* d_msg_bottom, float* d_msg_top);
int lbm_pair_finish(lbm_ctx* ctx);

/* Diagnostic of the two-steps-per-pass path (no counterpart in the reference): one k_fused2 launch from the current
 * state into the other population buffer in which every workgroup records its start and end time (100 MHz
 * device clock): stamps[2 i], stamps[2 i + 1] for workgroup i; *nworkgroups = workgroups of a launch (0: the fused
 * path is not in use).  stamps == NULL only asks for the count.  The state does not advance, but rho/u of the
 * last step can no longer be recomputed on demand until the next lbm_step().  tools/chain_spread.py prints how
 * evenly the block chains of a launch end. */
int lbm_fused_chain_stamps(lbm_ctx* ctx, uint64_t* stamps, size_t n, int* nworkgroups);

/* A halo link between two processes without RCCL (no counterpart in the reference, which has one device): what a
 * transport made of peer copies needs (cfd_amd/slab.py IpcCopyP2P).
 * lbm_ipc_alloc: `bytes` of zeroed memory on the current device -- fine-grained for flags, ordinary for message
 * buffers -- and (ipc_handle != NULL) the 64-byte IPC handle other processes map it with: lbm_ipc_open / lbm_ipc_close
 * there, lbm_ipc_free in the owner.  lbm_ipc_copy: an asynchronous device-to-device copy on `stream` (a hipStream_t),
 * e.g. into a neighbour's mapped buffer (a peer copy between GPUs).
 * Flags are 32-bit words of such a fine-grained block.  lbm_flag_post stores `value` into a flag -- its own or a
 * neighbour's -- when `stream` gets there; lbm_flag_await holds `stream` until the flag has reached `value` (compared as
 * a wrapping 32-bit sequence number), or until timeout_ms have passed: then it sets *d_timed_out (another word of the
 * rank's own block) to 1 and lets the stream go on, so a lost peer ends in an error the host can read (lbm_flag_peek,
 * blocking), not in a hung GPU; later waits that find the word set return at once.  No context argument: errors are
 * reported through lbm_last_error(NULL). */
int lbm_ipc_alloc(size_t bytes, int fine_grained, void** d_ptr, void* ipc_handle);
int lbm_ipc_free(void* d_ptr);
int lbm_ipc_open(const void* ipc_handle, void** d_ptr);
int lbm_ipc_close(void* d_ptr);
int lbm_ipc_copy(void* stream, void* d_dst, const void* d_src, size_t bytes);
int lbm_flag_post(void* stream, uint32_t* d_flag, uint32_t value);
int lbm_flag_await(void* stream, const uint32_t* d_flag, uint32_t value, uint32_t timeout_ms, uint32_t* d_timed_out);
/* up to LBM_FLAG_OPS_MAX posts, then up to LBM_FLAG_OPS_MAX waits (one time limit for all of them), in ONE launch: the
 * posts and waits of an exchange as IpcCopyP2P orders them (the host arrays are read before the call returns) */
#define LBM_FLAG_OPS_MAX 4
int lbm_flag_sync(void* stream, uint32_t* const* d_post, const uint32_t* post_values, int nposts,
                  const uint32_t* const* d_wait, const uint32_t* wait_values, int nwaits, uint32_t timeout_ms, uint32_t* d_timed_out);
int lbm_flag_peek(const uint32_t* d_flag, uint32_t* value);

#ifdef __cplusplus
}
#endif
#endif /* LBM_ABI_H */
