// lbm_solver.hpp -- the reference's solver interface on top of the C ABI (include/lbm_abi.h).
//
// Drop-in for Engine/src/Physics/lbm_solver.cuh:9-51 in a headless build of the engine: same
// class name, member names, argument meaning and error behaviour, so Engine/src/main.cpp:17-28,36
// compiles against it unchanged (minus the Vulkan interop calls).  Header-only; link liblbm_hip.so.
//
//   reference                                   here
//   ---------------------------------------     ------------------------------------------------
//   struct settings (lbm_solver.cuh:9-15)        same fields and defaults; read at every step()
//   cuda_math::vec4 (cuda_math.cuh:37-55)        same 16-byte {x,y,z,w}
//   cuda_buffer<T> (cuda_buffer.cuh:18-122)      hip_buffer<T>: the same members (release, memset, copy_from_host,
//                                                copy_to_host, copy_from_device, copy_to_device, get_data,
//                                                set_size) as views of the solver's fields in the reference
//                                                wire format
//   CUDA_CHECK -> log + exit (cuda_buffer.cuh:9) LBM_CHECK -> log + std::exit(EXIT_FAILURE)
//   register_external_* (lbm_solver.cu:558-717)  throw std::runtime_error (CUDA-Vulkan interop is
//                                                out of scope)
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/lbm_abi.h"

struct settings {
    float tau = 0.55f;  // dead in the reference's live kernel; kept for source compatibility
    float inlet_velocity = 0.08f;
    float omega_shear = 1.98f;
    float omega_bulk = 1.90f;
};

namespace cuda_math {
struct vec4 {
    float x, y, z, w;
};
}  // namespace cuda_math

#define LBM_CHECK(ctx, call)                                                                  \
    do {                                                                                      \
        int rc_ = (call);                                                                     \
        if (rc_ != LBM_OK) {                                                                  \
            std::fprintf(stderr, "[error] LBM error %d: %s at %s:%d\n", rc_, lbm_last_error(ctx), \
                         __FILE__, __LINE__);                                                 \
            std::exit(EXIT_FAILURE);                                                          \
        }                                                                                     \
    } while (0)

// View of one solver field with the call surface of cuda_buffer<T> that the engine uses.
template <typename T>
class hip_buffer {
   public:
    enum class field { density, velocity, solid, curl };
    hip_buffer() = default;
    hip_buffer(lbm_ctx* ctx, field f, size_t size) : m_ctx(ctx), m_field(f), m_size(size), m_full(size) {}
    hip_buffer(const hip_buffer&) = delete;  // reference quirk Q8 (shallow copy) not reproduced
    hip_buffer& operator=(const hip_buffer&) = delete;
    hip_buffer(hip_buffer&&) = default;
    hip_buffer& operator=(hip_buffer&&) = default;

    // cuda_buffer.cuh:79-84: blocking D2H copy of `size` elements, silently skipped if size > m_size
    void copy_to_host(T* host_data, size_t size) {
        if (!m_ctx || size > m_size || size > m_full) return;
        if (size == m_full) {
            read_all(host_data);
        } else if (size > 0) {   // a prefix: straight from the device array, no whole-field temporary
            const T* src = get_data();
            if (src) LBM_CHECK(m_ctx, lbm_copy_to_host(m_ctx, host_data, src, size * sizeof(T)));
        }
    }
    // cuda_buffer.cuh:93-98: blocking device-to-device copy of `size` elements out of the field (what a
    // renderer does with get_velocity() / get_solid_mask()); silently skipped if size > m_size
    void copy_to_device(T* device_ptr, size_t size) const {
        if (!m_ctx || size > m_size || size > m_full) return;
        const T* src = get_data();
        if (src) LBM_CHECK(m_ctx, lbm_device_copy(m_ctx, device_ptr, src, size * sizeof(T)));
    }
    // cuda_buffer.cuh:86-91.  The solid mask is an input of the solver and is replaced as a whole (lbm_solver.cu:555);
    // density / velocity / curl are outputs of step() in the reference: what is written into them shows until the
    // next step rewrites them and never reaches the populations -- the same here (lbm_write_field).
    void copy_from_device(const T* device_ptr, size_t size) {
        if (!m_ctx || size > m_size || size > m_full) return;
        write(device_ptr, size, 1);
    }
    // cuda_buffer.cuh:56-64: a released buffer answers every later call with a no-op / null.  The fields belong
    // to the solver (external, non-owning memory in the reference's terms, cuda_buffer.cuh:28-29), so nothing is freed.
    void release() {
        m_ctx = nullptr;
        m_size = 0;
        m_full = 0;
    }
    // cuda_buffer.cuh:116: changes the bound the copy calls check against, not the field
    void set_size(const size_t size) { m_size = size; }
    // cuda_buffer.cuh:72-77
    void copy_from_host(const T* host_data, size_t size) {
        if (!m_ctx || size > m_size || size > m_full) return;
        write(host_data, size, 0);
    }
    // cuda_buffer.cuh:66-70: byte-wise fill of the whole array (lbm_solver.cu:415: d_solid.memset(0))
    void memset(int value) {
        if (!m_ctx || m_size < m_full) return;
        std::vector<T> v(m_full);
        std::memset(static_cast<void*>(v.data()), value, m_full * sizeof(T));
        write(v.data(), m_full, 0);
    }
    // device pointer in the reference wire format, up to date and synchronised (cuda_buffer.cuh:100-114)
    T* get_data() { return const_cast<T*>(data_ptr()); }
    [[nodiscard]] const T* get_data() const { return data_ptr(); }
    size_t size() const { return m_size; }

   private:
    const T* data_ptr() const {
        if (!m_ctx) {
            std::fprintf(stderr, "[error] Trying to access null device pointer!\n");
            return nullptr;
        }
        float *rho = nullptr, *u = nullptr, *curl = nullptr;
        uint8_t* solid = nullptr;
        // only what this buffer stands for is brought up to date (the curl pass costs 20 B/cell)
        LBM_CHECK(m_ctx, lbm_device_fields(m_ctx, &rho, &u, m_field == field::curl ? &curl : nullptr,
                                           m_field == field::solid ? &solid : nullptr));
        switch (m_field) {
            case field::density: return reinterpret_cast<const T*>(rho);
            case field::velocity: return reinterpret_cast<const T*>(u);
            case field::curl: return reinterpret_cast<const T*>(curl);
            default: return reinterpret_cast<const T*>(solid);
        }
    }
    void write(const T* src, size_t size, int on_device) {
        if (m_field == field::solid) {
            if (size != m_full) {   // the packed mask is rebuilt as a whole
                std::fprintf(stderr, "[error] solid mask: only the whole mask can be written (%zu of %zu elements given)\n", size, m_full);
                return;
            }
            if (on_device) LBM_CHECK(m_ctx, lbm_set_solid_u8_device(m_ctx, reinterpret_cast<const uint8_t*>(src), size));
            else LBM_CHECK(m_ctx, lbm_set_solid_u8(m_ctx, reinterpret_cast<const uint8_t*>(src), size));
            return;
        }
        const int which = m_field == field::density ? LBM_FIELD_DENSITY : (m_field == field::velocity ? LBM_FIELD_VELOCITY : LBM_FIELD_CURL);
        LBM_CHECK(m_ctx, lbm_write_field(m_ctx, which, src, size, on_device));
    }
    void read_all(T* dst) {
        switch (m_field) {
            case field::density: LBM_CHECK(m_ctx, lbm_read_density(m_ctx, reinterpret_cast<float*>(dst), m_full)); break;
            case field::velocity: LBM_CHECK(m_ctx, lbm_read_velocity(m_ctx, reinterpret_cast<float*>(dst), m_full)); break;
            case field::curl: LBM_CHECK(m_ctx, lbm_read_curl(m_ctx, reinterpret_cast<float*>(dst), m_full)); break;
            case field::solid: LBM_CHECK(m_ctx, lbm_read_solid_u8(m_ctx, reinterpret_cast<uint8_t*>(dst), m_full)); break;
        }
    }
    lbm_ctx* m_ctx = nullptr;
    field m_field = field::density;
    size_t m_size = 0;   // the bound copy calls check against (set_size changes it)
    size_t m_full = 0;   // elements of the field itself
};

class lbm_solver {
   public:
    lbm_solver(int width, int height, int depth, int preset = LBM_PRESET_WIND_TUNNEL_REF,
               int scheme = LBM_SCHEME_AB)
        : m_width(width), m_height(height), m_depth(depth) {
        lbm_opts o{};
        o.struct_size = sizeof(lbm_opts);
        o.device = -1;
        o.preset = preset;
        o.scheme = scheme;
        o.plane_skew_bytes = -1;
        LBM_CHECK(nullptr, lbm_create(&m_ctx, width, height, depth, &o));
        const size_t n = static_cast<size_t>(width) * height * depth;
        d_rho = hip_buffer<float>(m_ctx, hip_buffer<float>::field::density, n);
        d_u = hip_buffer<cuda_math::vec4>(m_ctx, hip_buffer<cuda_math::vec4>::field::velocity, n);
        d_solid = hip_buffer<unsigned char>(m_ctx, hip_buffer<unsigned char>::field::solid, n);
        d_curl = hip_buffer<float>(m_ctx, hip_buffer<float>::field::curl, n);
    }
    ~lbm_solver() { lbm_destroy(m_ctx); }
    lbm_solver(const lbm_solver&) = delete;
    lbm_solver& operator=(const lbm_solver&) = delete;

    void init() { LBM_CHECK(m_ctx, lbm_init(m_ctx)); }
    void step() {  // lbm_solver.cu:477-494: settings are read at every launch
        LBM_CHECK(m_ctx, lbm_set_params(m_ctx, m_settings.inlet_velocity, m_settings.omega_shear, m_settings.omega_bulk));
        LBM_CHECK(m_ctx, lbm_step(m_ctx, 1));
    }
    void reset() { init(); }
    void compute_curl() { LBM_CHECK(m_ctx, lbm_compute_curl(m_ctx)); }

    struct Rect {
        int x, y, z, w, h, d;
    };
    void add_solid(const Rect& r) { LBM_CHECK(m_ctx, lbm_add_solid_rect(m_ctx, r.x, r.y, r.z, r.w, r.h, r.d)); }

    void register_external_density(int, size_t) { throw std::runtime_error("Failed to import external memory: Vulkan interop is not part of the HIP solver"); }
    void register_external_velocity(int, size_t) { throw std::runtime_error("Failed to import external velocity memory: Vulkan interop is not part of the HIP solver"); }
    void register_external_curl(int, size_t) { throw std::runtime_error("Failed to import external curl memory: Vulkan interop is not part of the HIP solver"); }
    void register_external_solid(int, size_t) { throw std::runtime_error("Failed to import external solid memory: Vulkan interop is not part of the HIP solver"); }

    void load_mesh(const std::string& path) {  // lbm_solver.cu:521-556: a missing mesh only logs
        const int rc = lbm_load_mesh(m_ctx, path.c_str());
        if (rc == LBM_ERR_IO) std::fprintf(stderr, "[error] Failed to load mesh: %s\n", path.c_str());
        else LBM_CHECK(m_ctx, rc);
    }

    hip_buffer<float>& get_density() { return d_rho; }
    hip_buffer<cuda_math::vec4>& get_velocity() { return d_u; }
    hip_buffer<unsigned char>& get_solid_mask() { return d_solid; }
    hip_buffer<float>& get_curl() { return d_curl; }
    // a `const lbm_solver&` consumer (a renderer that only reads) gets the const views
    const hip_buffer<float>& get_density() const { return d_rho; }
    const hip_buffer<cuda_math::vec4>& get_velocity() const { return d_u; }
    const hip_buffer<unsigned char>& get_solid_mask() const { return d_solid; }
    const hip_buffer<float>& get_curl() const { return d_curl; }

    int get_width() const { return m_width; }
    int get_height() const { return m_height; }
    int get_depth() const { return m_depth; }
    settings& get_settings() { return m_settings; }

    lbm_ctx* native_handle() { return m_ctx; }
    void sync() { LBM_CHECK(m_ctx, lbm_sync(m_ctx)); }

   private:
    int m_width, m_height, m_depth;
    settings m_settings;
    lbm_ctx* m_ctx = nullptr;
    hip_buffer<float> d_rho;
    hip_buffer<cuda_math::vec4> d_u;
    hip_buffer<unsigned char> d_solid;
    hip_buffer<float> d_curl;
};
