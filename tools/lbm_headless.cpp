// lbm_headless -- command-line caller of lbm_solver: what Engine/src/main.cpp's `cfd` layer does
// (construct 400x100x100, init, load_mesh, step per frame; main.cpp:11-39) without the window.
//   lbm_headless [--dims W H D] [--steps N] [--obj PATH] [--preset wind|cavity]
//                [--scheme ab|aa|slide] [--u U] [--omega-shear S] [--omega-bulk B] [--dump PREFIX]
// Prints MLUPS and field statistics; --dump writes PREFIX.{rho,u,solid,curl}.raw in the reference
// wire formats (idx = z*W*H + y*W + x) and PREFIX.json with the dimensions.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../cfd_amd/csrc/lbm_solver.hpp"

template <typename T>
static void dump(const std::string& path, const std::vector<T>& v) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { std::perror(path.c_str()); return; }
    std::fwrite(v.data(), sizeof(T), v.size(), f);
    std::fclose(f);
}

int main(int argc, char** argv) {
    int W = 400, H = 100, D = 100, steps = 100, preset = LBM_PRESET_WIND_TUNNEL_REF;  // main.cpp:12-14
    int scheme = LBM_SCHEME_AB;
    std::string obj, prefix;
    settings s;
    for (int i = 1; i < argc; i++) {
        auto is = [&](const char* a) { return !std::strcmp(argv[i], a); };
        if (is("--dims") && i + 3 < argc) { W = atoi(argv[++i]); H = atoi(argv[++i]); D = atoi(argv[++i]); }
        else if (is("--steps") && i + 1 < argc) steps = atoi(argv[++i]);
        else if (is("--obj") && i + 1 < argc) obj = argv[++i];
        else if (is("--dump") && i + 1 < argc) prefix = argv[++i];
        else if (is("--preset") && i + 1 < argc) preset = !std::strcmp(argv[++i], "cavity") ? LBM_PRESET_CAVITY : LBM_PRESET_WIND_TUNNEL_REF;
        else if (is("--scheme") && i + 1 < argc) { ++i; scheme = !std::strcmp(argv[i], "aa") ? LBM_SCHEME_AA : (!std::strcmp(argv[i], "slide") ? LBM_SCHEME_SLIDE : LBM_SCHEME_AB); }
        else if (is("--u") && i + 1 < argc) s.inlet_velocity = (float)atof(argv[++i]);
        else if (is("--omega-shear") && i + 1 < argc) s.omega_shear = (float)atof(argv[++i]);
        else if (is("--omega-bulk") && i + 1 < argc) s.omega_bulk = (float)atof(argv[++i]);
        else { std::fprintf(stderr, "usage: %s [--dims W H D] [--steps N] [--obj PATH] [--preset wind|cavity] [--scheme ab|aa|slide] [--u U] [--omega-shear S] [--omega-bulk B] [--dump PREFIX]\n", argv[0]); return 2; }
    }
    lbm_solver solver(W, H, D, preset, scheme);
    solver.get_settings() = s;
    solver.init();
    if (!obj.empty()) solver.load_mesh(obj);
    solver.step();  // warm-up launch (also builds the tile flags)
    solver.sync();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i < steps; i++) solver.step();
    solver.sync();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const size_t n = (size_t)W * H * D;
    std::vector<float> rho(n), curl(n);
    std::vector<cuda_math::vec4> u(n);
    std::vector<unsigned char> solid(n);
    solver.get_density().copy_to_host(rho.data(), n);
    solver.get_velocity().copy_to_host(u.data(), n);
    solver.get_solid_mask().copy_to_host(solid.data(), n);
    solver.get_curl().copy_to_host(curl.data(), n);
    double sum = 0, umax = 0, cmax = 0;
    size_t nsolid = 0;
    for (size_t i = 0; i < n; i++) {
        sum += rho[i];
        umax = std::fmax(umax, std::sqrt((double)u[i].x * u[i].x + (double)u[i].y * u[i].y + (double)u[i].z * u[i].z));
        cmax = std::fmax(cmax, (double)curl[i]);
        nsolid += solid[i] != 0;
    }
    std::printf("%dx%dx%d steps=%d solid=%zu sum_rho=%.9f umax=%.9g curlmax=%.9g", W, H, D, steps, nsolid, sum, umax, cmax);
    if (steps > 1) std::printf(" MLUPS=%.1f", (double)n * (steps - 1) / dt / 1e6);
    std::printf("\n");
    if (!prefix.empty()) {
        dump(prefix + ".rho.raw", rho); dump(prefix + ".u.raw", u); dump(prefix + ".solid.raw", solid); dump(prefix + ".curl.raw", curl);
        FILE* f = std::fopen((prefix + ".json").c_str(), "w");
        if (f) { std::fprintf(f, "{\"width\": %d, \"height\": %d, \"depth\": %d, \"steps\": %d, \"layout\": \"idx = z*W*H + y*W + x; u = float4 (x,y,z,0)\"}\n", W, H, D, steps); std::fclose(f); }
    }
    return 0;
}
