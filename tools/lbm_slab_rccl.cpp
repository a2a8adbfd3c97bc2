// lbm_slab_rccl -- native multi-GPU driver: one process, one z-slab per GPU, halo exchange by RCCL
// ncclSend/ncclRecv (over xGMI between the GPUs of a node), all host code C++ on the C ABI
// (include/lbm_abi.h).  The reference is single-GPU (no NCCL/MPI call sites); this is
// the C++ twin of cfd_amd/slab.py, which drives the same ABI from one process per GPU.
//
//   lbm_slab_rccl [--dims W H D] [--gpus P] [--steps N] [--obj PATH] [--preset wind|cavity]
//                 [--scheme ab|aa|slide] [--autotune] [--single-steps] [--copy-engine] [--check]
//
// Per step and slab (two streams):
//   side stream : lbm_step_boundary_layers (writes both outgoing messages) -> ncclSend/ncclRecv
//   main stream : lbm_step_interior_layers, meanwhile
//   main stream : waits for the exchange, lbm_halo_unpack_both, lbm_step_finish
// Where every slab offers it (A/B, two steps per pass: lbm_pair_available) all but the last step run as PAIRS:
//   side stream : lbm_pair_boundary_first -> exchange -> lbm_halo_unpack_both -> lbm_pair_boundary_second
//                 -> exchange -> lbm_halo_unpack_both
//   main stream : lbm_pair_interior (the fused kernel advances the interior by two steps meanwhile)
//   main stream : waits for the side stream, lbm_pair_finish
// (--single-steps keeps every step a single split step.)
// The wind-tunnel preset exchanges around a ring (the reference's unchecked outlet index wraps from
// the last z layer to the first, quirk Q2), the cavity along an open chain.
// --check runs the undecomposed domain on GPU 0 as well and compares density and velocity bit for bit.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/lbm_abi.h"

#define HIP_OK(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } \
    } while (0)
#define NCCL_OK(call)                                                                             \
    do {                                                                                          \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s (%s:%d)\n", #call, ncclGetErrorString(r_), __FILE__, __LINE__); std::exit(1); } \
    } while (0)
#define LBM_OKAY(ctx, call)                                                                       \
    do {                                                                                          \
        int rc_ = (call);                                                                         \
        if (rc_ != LBM_OK) { std::fprintf(stderr, "%s: %s (%s:%d)\n", #call, lbm_last_error(ctx), __FILE__, __LINE__); std::exit(1); } \
    } while (0)

struct Slab {
    int dev = 0, z0 = 0, depth = 0, down = -1, up = -1;
    lbm_ctx* ctx = nullptr;
    hipStream_t main = nullptr, side = nullptr;
    hipEvent_t main_done = nullptr, side_done = nullptr;
    hipEvent_t ready = nullptr, sent = nullptr;   // --copy-engine: this slab's earlier work is done / its messages have been delivered
    float* send[2] = {nullptr, nullptr};  // [0] leaves through the bottom face, [1] through the top face
    float* recv[2] = {nullptr, nullptr};  // [0] arrived on the bottom face, [1] on the top face
    ncclComm_t comm = nullptr;
};

// --copy-engine: the same exchange as peer-to-peer copies (hipMemcpyPeerAsync: the SDMA engines over xGMI, no compute
// units) instead of RCCL's send / recv kernels, which find no room beside the two-steps-per-pass kernel and wait for its
// launch to end (profiles/r03_tuning.md section 8).  One process holds every GPU, so a sender can write its neighbour's
// receive buffer directly; two events per slab order it: a slab's copies wait until the neighbour has got as far as this
// exchange (its earlier unpack has consumed the buffer), and a slab goes on once its neighbours' copies have landed.
static bool g_copy_engine = false;
static void exchange_copies(std::vector<Slab>& slabs, size_t nfloats, bool on_side) {
    const size_t bytes = nfloats * sizeof(float);
    auto stream_of = [&](Slab& s) { return on_side ? s.side : s.main; };
    for (Slab& s : slabs) { HIP_OK(hipSetDevice(s.dev)); HIP_OK(hipEventRecord(s.ready, stream_of(s))); }
    for (Slab& s : slabs) {
        HIP_OK(hipSetDevice(s.dev));
        hipStream_t st = stream_of(s);
        if (s.up >= 0) {     // leaves through the top face, arrives on the neighbour's bottom face
            HIP_OK(hipStreamWaitEvent(st, slabs[(size_t)s.up].ready, 0));
            HIP_OK(hipMemcpyPeerAsync(slabs[(size_t)s.up].recv[0], slabs[(size_t)s.up].dev, s.send[1], s.dev, bytes, st));
        }
        if (s.down >= 0) {
            HIP_OK(hipStreamWaitEvent(st, slabs[(size_t)s.down].ready, 0));
            HIP_OK(hipMemcpyPeerAsync(slabs[(size_t)s.down].recv[1], slabs[(size_t)s.down].dev, s.send[0], s.dev, bytes, st));
        }
        HIP_OK(hipEventRecord(s.sent, st));
    }
    for (Slab& s : slabs) {
        HIP_OK(hipSetDevice(s.dev));
        if (s.down >= 0) HIP_OK(hipStreamWaitEvent(stream_of(s), slabs[(size_t)s.down].sent, 0));
        if (s.up >= 0) HIP_OK(hipStreamWaitEvent(stream_of(s), slabs[(size_t)s.up].sent, 0));
    }
}

static void exchange(std::vector<Slab>& slabs, size_t nfloats, bool on_side) {
    if (g_copy_engine) { exchange_copies(slabs, nfloats, on_side); return; }
    NCCL_OK(ncclGroupStart());
    for (Slab& s : slabs) {
        HIP_OK(hipSetDevice(s.dev));
        hipStream_t st = on_side ? s.side : s.main;
        if (s.up >= 0) NCCL_OK(ncclSend(s.send[1], nfloats, ncclFloat, s.up, s.comm, st));
        if (s.down >= 0) NCCL_OK(ncclSend(s.send[0], nfloats, ncclFloat, s.down, s.comm, st));
        if (s.down >= 0) NCCL_OK(ncclRecv(s.recv[0], nfloats, ncclFloat, s.down, s.comm, st));
        if (s.up >= 0) NCCL_OK(ncclRecv(s.recv[1], nfloats, ncclFloat, s.up, s.comm, st));
    }
    NCCL_OK(ncclGroupEnd());
}

int main(int argc, char** argv) {
    int W = 256, H = 256, D = 256, P = 0, steps = 100, preset = LBM_PRESET_WIND_TUNNEL_REF, scheme = LBM_SCHEME_AB;
    bool check = false, autotune = false, single_steps = false;
    std::string obj;
    for (int i = 1; i < argc; i++) {
        auto is = [&](const char* a) { return !std::strcmp(argv[i], a); };
        if (is("--dims") && i + 3 < argc) { W = atoi(argv[++i]); H = atoi(argv[++i]); D = atoi(argv[++i]); }
        else if (is("--gpus") && i + 1 < argc) P = atoi(argv[++i]);
        else if (is("--steps") && i + 1 < argc) steps = atoi(argv[++i]);
        else if (is("--obj") && i + 1 < argc) obj = argv[++i];
        else if (is("--preset") && i + 1 < argc) preset = !std::strcmp(argv[++i], "cavity") ? LBM_PRESET_CAVITY : LBM_PRESET_WIND_TUNNEL_REF;
        else if (is("--scheme") && i + 1 < argc) { ++i; scheme = !std::strcmp(argv[i], "aa") ? LBM_SCHEME_AA : (!std::strcmp(argv[i], "slide") ? LBM_SCHEME_SLIDE : LBM_SCHEME_AB); }
        else if (is("--check")) check = true;
        else if (is("--autotune")) autotune = true;
        else if (is("--single-steps")) single_steps = true;
        else if (is("--copy-engine")) g_copy_engine = true;
        else { std::fprintf(stderr, "usage: %s [--dims W H D] [--gpus P] [--steps N] [--obj PATH] [--preset wind|cavity] [--scheme ab|aa|slide] [--autotune] [--single-steps] [--copy-engine] [--check]\n", argv[0]); return 2; }
    }
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    if (P <= 0) P = ndev;
    if (P > ndev || P > D) { std::fprintf(stderr, "%d slabs need %d GPUs (have %d) and at least %d layers\n", P, P, ndev, P); return 2; }

    std::vector<Slab> slabs(P);
    std::vector<ncclComm_t> comms(P);
    std::vector<int> devs(P);
    for (int r = 0; r < P; r++) devs[r] = r;
    NCCL_OK(ncclCommInitAll(comms.data(), P, devs.data()));
    const bool ring = preset == LBM_PRESET_WIND_TUNNEL_REF;
    size_t nfloats = 0;
    for (int r = 0; r < P; r++) {
        Slab& s = slabs[r];
        s.dev = r; s.comm = comms[r];
        s.z0 = (int)((long long)D * r / P);
        s.depth = (int)((long long)D * (r + 1) / P) - s.z0;
        s.down = r > 0 ? r - 1 : (ring ? P - 1 : -1);
        s.up = r < P - 1 ? r + 1 : (ring ? 0 : -1);
        HIP_OK(hipSetDevice(r));
        HIP_OK(hipStreamCreateWithFlags(&s.main, hipStreamNonBlocking));
        HIP_OK(hipStreamCreateWithFlags(&s.side, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&s.main_done, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&s.side_done, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&s.ready, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&s.sent, hipEventDisableTiming));
        if (g_copy_engine)
            for (int q = 0; q < P; q++)
                if (q != r) { const hipError_t e = hipDeviceEnablePeerAccess(q, 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_OK(e); (void)hipGetLastError(); }
        lbm_opts o;
        std::memset(&o, 0, sizeof o);
        o.struct_size = sizeof o; o.device = r; o.preset = preset; o.scheme = scheme;
        o.z0 = s.z0; o.depth_global = D; o.stream = s.main; o.boundary_stream = s.side; o.plane_skew_bytes = -1;
        LBM_OKAY(nullptr, lbm_create(&s.ctx, W, H, s.depth, &o));
        if (scheme != LBM_SCHEME_AB && !check) LBM_OKAY(s.ctx, lbm_set_field_window(s.ctx, 0, 0));  // no per-step rho/u
        LBM_OKAY(s.ctx, lbm_halo_message_floats(s.ctx, &nfloats));
        for (int k = 0; k < 2; k++) {
            HIP_OK(hipMalloc(&s.send[k], nfloats * sizeof(float)));
            HIP_OK(hipMalloc(&s.recv[k], nfloats * sizeof(float)));
            HIP_OK(hipMemset(s.send[k], 0, nfloats * sizeof(float)));
            HIP_OK(hipMemset(s.recv[k], 0, nfloats * sizeof(float)));
        }
        LBM_OKAY(s.ctx, lbm_init(s.ctx));
        if (!obj.empty()) {
            const int rc = lbm_load_mesh(s.ctx, obj.c_str());  // every slab voxelises its own layers on its GPU
            if (rc == LBM_ERR_IO) std::fprintf(stderr, "[error] Failed to load mesh: %s\n", obj.c_str());
            else LBM_OKAY(s.ctx, rc);
        }
    }
    // static ghost layers of the solid mask (staged through the host once; the ring link carries none)
    {
        std::vector<std::vector<uint8_t>> lay(2 * P, std::vector<uint8_t>((size_t)W * H));
        for (int r = 0; r < P; r++)
            for (int side = 0; side < 2; side++) LBM_OKAY(slabs[r].ctx, lbm_halo_get_mask_layer(slabs[r].ctx, side, lay[2 * r + side].data(), lay[2 * r + side].size()));
        for (int r = 0; r < P; r++) {
            if (r > 0) LBM_OKAY(slabs[r].ctx, lbm_halo_set_mask_ghost(slabs[r].ctx, 0, lay[2 * (r - 1) + 1].data(), (size_t)W * H));
            if (r < P - 1) LBM_OKAY(slabs[r].ctx, lbm_halo_set_mask_ghost(slabs[r].ctx, 1, lay[2 * (r + 1)].data(), (size_t)W * H));
        }
    }
    // ghost layers of the populations from the initial state
    for (Slab& s : slabs) {
        LBM_OKAY(s.ctx, lbm_halo_pack(s.ctx, 0, s.send[0]));
        LBM_OKAY(s.ctx, lbm_halo_pack(s.ctx, 1, s.send[1]));
    }
    exchange(slabs, nfloats, false);
    for (Slab& s : slabs) {
        if (s.down >= 0) LBM_OKAY(s.ctx, lbm_halo_unpack(s.ctx, 0, s.recv[0]));
        if (s.up >= 0) LBM_OKAY(s.ctx, lbm_halo_unpack(s.ctx, 1, s.recv[1]));
        HIP_OK(hipSetDevice(s.dev));
        HIP_OK(hipEventRecord(s.main_done, s.main));
    }

    auto step_all = [&](int n) {
        for (int it = 0; it < n; it++) {
            for (Slab& s : slabs) {
                HIP_OK(hipSetDevice(s.dev));
                HIP_OK(hipStreamWaitEvent(s.side, s.main_done, 0));  // previous unpack / swap is done
                LBM_OKAY(s.ctx, lbm_step_boundary_layers(s.ctx, s.down >= 0 ? s.send[0] : nullptr, s.up >= 0 ? s.send[1] : nullptr));
            }
            for (Slab& s : slabs) LBM_OKAY(s.ctx, lbm_step_interior_layers(s.ctx));  // issued before the RCCL calls
            exchange(slabs, nfloats, true);
            for (Slab& s : slabs) {
                HIP_OK(hipSetDevice(s.dev));
                HIP_OK(hipEventRecord(s.side_done, s.side));
                HIP_OK(hipStreamWaitEvent(s.main, s.side_done, 0));
                LBM_OKAY(s.ctx, lbm_halo_unpack_both(s.ctx, s.down >= 0 ? s.recv[0] : nullptr, s.up >= 0 ? s.recv[1] : nullptr));
                LBM_OKAY(s.ctx, lbm_step_finish(s.ctx));
                HIP_OK(hipEventRecord(s.main_done, s.main));
            }
        }
    };
    // two steps per exchange pattern (cfd_amd/slab.py SlabSolver._step is the same schedule from Python)
    auto pair_all = [&](int npairs) {
        for (int it = 0; it < npairs; it++) {
            for (Slab& s : slabs) {
                HIP_OK(hipSetDevice(s.dev));
                HIP_OK(hipStreamWaitEvent(s.side, s.main_done, 0));
                LBM_OKAY(s.ctx, lbm_pair_boundary_first(s.ctx, s.down >= 0 ? s.send[0] : nullptr, s.up >= 0 ? s.send[1] : nullptr));
            }
            for (Slab& s : slabs) LBM_OKAY(s.ctx, lbm_pair_interior(s.ctx));
            exchange(slabs, nfloats, true);
            for (Slab& s : slabs) {   // the neighbours' t+1 layers, then the boundary layers' second step: side stream
                LBM_OKAY(s.ctx, lbm_halo_unpack_both(s.ctx, s.down >= 0 ? s.recv[0] : nullptr, s.up >= 0 ? s.recv[1] : nullptr));
                LBM_OKAY(s.ctx, lbm_pair_boundary_second(s.ctx, s.down >= 0 ? s.send[0] : nullptr, s.up >= 0 ? s.send[1] : nullptr));
            }
            exchange(slabs, nfloats, true);
            for (Slab& s : slabs) {
                HIP_OK(hipSetDevice(s.dev));
                LBM_OKAY(s.ctx, lbm_halo_unpack_both(s.ctx, s.down >= 0 ? s.recv[0] : nullptr, s.up >= 0 ? s.recv[1] : nullptr));
                HIP_OK(hipEventRecord(s.side_done, s.side));
                HIP_OK(hipStreamWaitEvent(s.main, s.side_done, 0));
                LBM_OKAY(s.ctx, lbm_pair_finish(s.ctx));
                HIP_OK(hipEventRecord(s.main_done, s.main));
            }
        }
    };
    auto pairs_offered = [&] {
        if (single_steps) return false;
        for (Slab& s : slabs) {
            int a = 0;
            LBM_OKAY(s.ctx, lbm_pair_available(s.ctx, &a));
            if (!a) return false;
        }
        return true;
    };
    // n steps: pairs for all but the last one or two (the last step of a batch stays a single step: rho/u on demand)
    auto advance = [&](int n) {
        const int pairs = (n >= 3 && pairs_offered()) ? (n - 1) / 2 : 0;
        pair_all(pairs);
        step_all(n - 2 * pairs);
        return pairs;
    };
    auto sync_all = [&] {
        for (Slab& s : slabs) { HIP_OK(hipSetDevice(s.dev)); HIP_OK(hipDeviceSynchronize()); }
    };

    // HIP maps streams onto a few hardware queues in creation order and streams that share a queue
    // serialise: time a few ordinary steps on every ordered pair of four fresh
    // streams per GPU and keep the fastest pair.
    int tuned = 0;
    if (autotune) {
        const int K = 4, trial = 6;
        std::vector<std::vector<hipStream_t>> pool(P, std::vector<hipStream_t>(K));
        for (int r = 0; r < P; r++) {
            HIP_OK(hipSetDevice(r));
            for (int k = 0; k < K; k++) HIP_OK(hipStreamCreateWithFlags(&pool[r][k], hipStreamNonBlocking));
        }
        auto bind = [&](int i, int j) {
            sync_all();
            for (int r = 0; r < P; r++) {
                LBM_OKAY(slabs[r].ctx, lbm_set_streams(slabs[r].ctx, pool[r][i], pool[r][j]));
                slabs[r].main = pool[r][i]; slabs[r].side = pool[r][j];
                HIP_OK(hipSetDevice(r));
                HIP_OK(hipEventRecord(slabs[r].main_done, slabs[r].main));
            }
        };
        double best = 1e30;
        int bi = 0, bj = 1;
        for (int i = 0; i < K; i++)
            for (int j = 0; j < K; j++) {
                if (i == j) continue;
                bind(i, j);
                step_all(2);
                sync_all();
                const auto a = std::chrono::steady_clock::now();
                step_all(trial);
                sync_all();
                const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() / trial;
                tuned += 2 + trial;
                if (t < best) { best = t; bi = i; bj = j; }
            }
        bind(bi, bj);
        std::printf("stream autotune: pair (%d, %d) of %d, %.3f ms/step in the trial\n", bi, bj, K, best * 1e3);
    }
    const int warm = steps - tuned > 20 ? 10 : 0;
    steps -= tuned;  // the trials were ordinary steps of the simulation
    if (steps < warm) steps = warm;
    advance(warm);
    sync_all();
    const auto t0 = std::chrono::steady_clock::now();
    const int paired = advance(steps - warm);
    sync_all();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double cells = (double)W * H * D;
    std::printf("%dx%dx%d in %d z-slabs over %s (%s, %s): %d steps (%d pairs of steps in the timed part), %.3f ms/step, %.1f MLUPS\n", W, H, D, P, g_copy_engine ? "peer copies" : "RCCL",
                ring ? "wind tunnel, ring" : "cavity, chain", scheme == LBM_SCHEME_AA ? "AA" : (scheme == LBM_SCHEME_SLIDE ? "sliding A/B" : "A/B"), steps + tuned,
                paired, dt / (steps - warm) * 1e3, cells * (steps - warm) / dt / 1e6);

    int bad = 0;
    if (check) {
        HIP_OK(hipSetDevice(0));
        lbm_ctx* ref = nullptr;
        lbm_opts o;
        std::memset(&o, 0, sizeof o);
        o.struct_size = sizeof o; o.device = 0; o.preset = preset; o.scheme = LBM_SCHEME_AB; o.plane_skew_bytes = -1;
        LBM_OKAY(nullptr, lbm_create(&ref, W, H, D, &o));
        LBM_OKAY(ref, lbm_init(ref));
        if (!obj.empty()) (void)lbm_load_mesh(ref, obj.c_str());
        LBM_OKAY(ref, lbm_step(ref, steps + tuned));
        const size_t n = (size_t)W * H * D;
        std::vector<float> rho_ref(n), u_ref(4 * n), rho(n), u(4 * n);
        LBM_OKAY(ref, lbm_read_density(ref, rho_ref.data(), n));
        LBM_OKAY(ref, lbm_read_velocity(ref, u_ref.data(), n));
        lbm_destroy(ref);
        for (Slab& s : slabs) {
            const size_t off = (size_t)s.z0 * W * H, cnt = (size_t)s.depth * W * H;
            LBM_OKAY(s.ctx, lbm_read_density(s.ctx, rho.data() + off, cnt));
            LBM_OKAY(s.ctx, lbm_read_velocity(s.ctx, u.data() + 4 * off, cnt));
        }
        const bool same = !std::memcmp(rho.data(), rho_ref.data(), n * sizeof(float)) && !std::memcmp(u.data(), u_ref.data(), 4 * n * sizeof(float));
        std::printf("check against the undecomposed run on GPU 0: %s\n", same ? "bit-identical" : "DIFFERENT");
        bad = same ? 0 : 1;
    }
    for (Slab& s : slabs) {
        HIP_OK(hipSetDevice(s.dev));
        lbm_destroy(s.ctx);
        for (int k = 0; k < 2; k++) { (void)hipFree(s.send[k]); (void)hipFree(s.recv[k]); }
        (void)hipEventDestroy(s.main_done); (void)hipEventDestroy(s.side_done);
        (void)hipStreamDestroy(s.main); (void)hipStreamDestroy(s.side);
        ncclCommDestroy(s.comm);
    }
    return bad;
}
