// C++ tests of the lbm_solver mirror, written after the reference's own
// Engine/tests/test_lbm_solver.cu:7-57 (Initialization, ResetAndStep), adapted to the 3-argument
// constructor the reference class actually has, with plain checks instead of gtest (not in the
// image).  The reference's "+-1.0 mass" expectation cannot hold for its open wind tunnel
// (SURVEY.md section 4), so ResetAndStep checks the survey's known answers instead.
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../cfd_amd/csrc/lbm_solver.hpp"

static int failures = 0;
#define EXPECT(cond)                                                    \
    do {                                                                \
        if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

static void Initialization() {
    const int width = 128, height = 64, depth = 32;
    lbm_solver solver(width, height, depth);
    solver.init();
    std::vector<float> host_rho(static_cast<size_t>(width) * height * depth, -1.0f);
    solver.get_density().copy_to_host(host_rho.data(), host_rho.size());
    bool all_one = true;
    for (float r : host_rho) all_one = all_one && r == 1.0f;
    EXPECT(all_one);
    EXPECT(solver.get_width() == width && solver.get_height() == height && solver.get_depth() == depth);
    // cuda_buffer::copy_to_host silently skips oversized requests (cuda_buffer.cuh:79-84)
    std::vector<float> big(host_rho.size() + 1, -7.0f);
    solver.get_density().copy_to_host(big.data(), big.size());
    EXPECT(big[0] == -7.0f);
    // partial copies take the leading elements
    std::vector<float> part(10, -7.0f);
    solver.get_density().copy_to_host(part.data(), part.size());
    EXPECT(part[9] == 1.0f);
}

static void ResetAndStep(const char* cube_obj) {
    lbm_solver solver(64, 64, 64);
    solver.init();
    solver.reset();
    solver.load_mesh(cube_obj);
    const size_t n = 64 * 64 * 64;
    std::vector<unsigned char> mask(n);
    solver.get_solid_mask().copy_to_host(mask.data(), n);
    size_t solid = 0;
    for (unsigned char m : mask) solid += m != 0;
    EXPECT(solid == 9196);
    for (int i = 0; i < 10; i++) solver.step();
    std::vector<float> rho(n);
    solver.get_density().copy_to_host(rho.data(), n);
    double sum = 0;
    for (float r : rho) sum += r;
    EXPECT(std::isfinite(sum));
    EXPECT(std::fabs(sum - 255103.576208234) < 1e-6);
    std::vector<cuda_math::vec4> u(n);
    solver.get_velocity().copy_to_host(u.data(), n);
    double umax = 0;
    for (const auto& v : u) {
        umax = std::fmax(umax, std::sqrt((double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z));
        if (v.w != 0.0f) failures++;
    }
    EXPECT(std::fabs(umax - 0.0652680206) < 1e-9);
    // live settings: a change between steps takes effect at the next step
    solver.get_settings().inlet_velocity = 0.05f;
    solver.step();
    EXPECT(solver.get_density().get_data() != nullptr);
    // missing mesh: logs and carries on with the floor-only mask (lbm_solver.cu:522-524)
    solver.load_mesh("/nonexistent.obj");
    solver.get_solid_mask().copy_to_host(mask.data(), n);
    solid = 0;
    for (unsigned char m : mask) solid += m != 0;
    EXPECT(solid == 64 * 64);
    bool threw = false;
    try { solver.register_external_velocity(3, 16); } catch (const std::runtime_error&) { threw = true; }
    EXPECT(threw);
}

// cuda_buffer surface on the solid mask (copy_from_host, memset) and the in-place scheme behind
// the same class: AA and the sliding scheme must give the A/B fields bit for bit.
static void MaskWritesAndSchemes() {
    const int W = 96, H = 32, D = 24;
    const size_t n = static_cast<size_t>(W) * H * D;
    std::vector<unsigned char> mask(n, 0);
    for (int z = 8; z < 16; z++)
        for (int y = 0; y < 12; y++)
            for (int x = 30; x < 40; x++) mask[(static_cast<size_t>(z) * H + y) * W + x] = 255;
    std::vector<float> rho[3];
    for (int scheme = 0; scheme < 3; scheme++) {
        lbm_solver solver(W, H, D, LBM_PRESET_WIND_TUNNEL_REF, scheme);  // LBM_SCHEME_AB, _AA, _SLIDE
        solver.init();
        solver.add_solid({1, 1, 1, 4, 4, 4});
        solver.get_solid_mask().memset(0);                       // lbm_solver.cu:415
        solver.get_solid_mask().copy_from_host(mask.data(), n);  // lbm_solver.cu:555
        std::vector<unsigned char> back(n, 7);
        solver.get_solid_mask().copy_to_host(back.data(), n);
        EXPECT(back == mask);
        for (int i = 0; i < 7; i++) solver.step();
        rho[scheme].resize(n);
        solver.get_density().copy_to_host(rho[scheme].data(), n);
    }
    EXPECT(rho[0] == rho[1]);
    EXPECT(rho[0] == rho[2]);
}

// the rest of cuda_buffer's surface (cuda_buffer.cuh:56-64, 86-98, 116): device-to-device copies out of a field
// and into the mask, set_size, release.  The "renderer side" device arrays here are other solvers' fields.
// what a read-only consumer of the solver compiles against (cuda_buffer.cuh:93,108-114: const overloads)
static void read_only_consumer(const lbm_solver& s, std::vector<float>& rho_prefix, cuda_math::vec4* d_dst, size_t n) {
    const float* d_rho = s.get_density().get_data();
    EXPECT(d_rho != nullptr);
    s.get_velocity().copy_to_device(d_dst, n);
    EXPECT(s.get_density().size() == n && rho_prefix.size() <= n);
}

// cuda_buffer<T>::copy_from_host / memset on the OUTPUT arrays (cuda_buffer.cuh:66-77): the reference shows what was
// written until the next step rewrites the array, and the populations never see it
static void FieldWritesAndConstViews() {
    const int W = 72, H = 18, D = 10;
    const size_t n = static_cast<size_t>(W) * H * D;
    lbm_solver a(W, H, D), b(W, H, D);
    a.init(); b.init();
    a.add_solid({20, 0, 3, 6, 5, 4}); b.add_solid({20, 0, 3, 6, 5, 4});
    for (int i = 0; i < 4; i++) { a.step(); b.step(); }
    std::vector<float> seed(n, 2.5f), got(n, 0.0f), rb(n), ra(n);
    a.get_density().copy_from_host(seed.data(), n);             // a test harness seeding rho
    a.get_density().copy_to_host(got.data(), n);
    EXPECT(got == seed);
    std::vector<float> part(7, -3.0f);
    a.get_density().copy_from_host(part.data(), 7);             // a prefix
    a.get_density().copy_to_host(got.data(), 9);                // a prefix back, without a whole-field temporary
    EXPECT(got[0] == -3.0f && got[6] == -3.0f && got[7] == 2.5f && got[8] == 2.5f);
    a.get_velocity().memset(0);
    std::vector<cuda_math::vec4> ua(n), ub(n);
    a.get_velocity().copy_to_host(ua.data(), n);
    bool zero = true;
    for (const auto& v : ua) zero = zero && v.x == 0.0f && v.y == 0.0f && v.z == 0.0f && v.w == 0.0f;
    EXPECT(zero);
    a.get_curl().memset(0);                                     // the curl array likewise
    a.get_curl().copy_to_host(got.data(), n);
    EXPECT(got[n / 2] == 0.0f);
    // the next step rewrites all of it: a and b still agree bit for bit
    a.step(); b.step();
    a.get_density().copy_to_host(ra.data(), n);
    b.get_density().copy_to_host(rb.data(), n);
    EXPECT(ra == rb);
    a.get_velocity().copy_to_host(ua.data(), n);
    b.get_velocity().copy_to_host(ub.data(), n);
    EXPECT(std::memcmp(ua.data(), ub.data(), n * sizeof(cuda_math::vec4)) == 0);
    // a partial write of the solid mask is refused (the packed mask is rebuilt as a whole), a whole one taken
    std::vector<unsigned char> m(n, 0), m2(n, 1);
    a.get_solid_mask().copy_from_host(m.data(), 5);
    a.get_solid_mask().copy_to_host(m2.data(), n);
    b.get_solid_mask().copy_to_host(m.data(), n);
    EXPECT(m == m2);
    // const views
    std::vector<float> pre(4);
    read_only_consumer(a, pre, b.get_velocity().get_data(), n);
    b.get_velocity().copy_to_host(ub.data(), n);
    EXPECT(std::memcmp(ua.data(), ub.data(), n * sizeof(cuda_math::vec4)) == 0);
}

static void DeviceCopiesSetSizeRelease() {
    const int W = 80, H = 20, D = 12;
    const size_t n = static_cast<size_t>(W) * H * D;
    lbm_solver a(W, H, D), b(W, H, D);
    a.init(); b.init();
    a.add_solid({10, 0, 2, 9, 7, 5});
    for (int i = 0; i < 5; i++) { a.step(); b.step(); }
    // b's mask := a's mask, device to device; afterwards both must evolve alike once b restarts
    b.get_solid_mask().copy_from_device(a.get_solid_mask().get_data(), n);
    std::vector<unsigned char> ma(n), mb(n, 9);
    a.get_solid_mask().copy_to_host(ma.data(), n);
    b.get_solid_mask().copy_to_host(mb.data(), n);
    EXPECT(ma == mb);
    a.reset(); b.reset();
    for (int i = 0; i < 6; i++) { a.step(); b.step(); }
    // a's velocity -> b's velocity array on the device (the copy a renderer makes), read back through b
    b.step();                                                   // b is one step ahead: the fields differ
    std::vector<cuda_math::vec4> ua(n), ub(n);
    a.get_velocity().copy_to_host(ua.data(), n);
    b.get_velocity().copy_to_host(ub.data(), n);
    EXPECT(std::memcmp(ua.data(), ub.data(), n * sizeof(cuda_math::vec4)) != 0);
    cuda_math::vec4* dst = b.get_velocity().get_data();
    a.get_velocity().copy_to_device(dst, n);
    b.get_velocity().copy_to_host(ub.data(), n);               // b's device array now holds a's field
    EXPECT(std::memcmp(ua.data(), ub.data(), n * sizeof(cuda_math::vec4)) == 0);
    // oversized device copies are skipped silently like the host ones
    a.get_density().copy_to_device(b.get_density().get_data(), n + 1);
    // set_size lowers the bound the copies check
    std::vector<float> r(n, -1.0f);
    a.get_density().set_size(10);
    a.get_density().copy_to_host(r.data(), n);                  // n > 10: skipped
    EXPECT(r[0] == -1.0f);
    a.get_density().copy_to_host(r.data(), 10);
    EXPECT(r[0] > 0.0f && r[9] > 0.0f && r[10] == -1.0f);
    a.get_density().set_size(n);
    // a released view answers nothing any more
    a.get_curl().release();
    EXPECT(a.get_curl().size() == 0);
    std::vector<float> c(n, -2.0f);
    a.get_curl().copy_to_host(c.data(), n);
    EXPECT(c[0] == -2.0f);
}

int main(int argc, char** argv) {
    Initialization();
    ResetAndStep(argc > 1 ? argv[1] : "tests/golden/cube.obj");
    MaskWritesAndSchemes();
    DeviceCopiesSetSizeRelease();
    FieldWritesAndConstViews();
    std::printf(failures ? "%d FAILED\n" : "all C++ lbm_solver tests passed\n", failures);
    return failures ? 1 : 0;
}
