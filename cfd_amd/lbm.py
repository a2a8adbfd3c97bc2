"""ctypes binding of liblbm_hip.so (include/lbm_abi.h) and a Python mirror of the reference's
`lbm_solver` class (Engine/src/Physics/lbm_solver.cuh:17-51).

This is host-side plumbing for tests and bench.py; the C++ mirror a maintainer would compile into
the engine is cfd_amd/csrc/lbm_solver.hpp.  There is no CPU path here: without the HIP library or
without a GPU every compute call raises.
"""
import ctypes as C
import os

import numpy as np

Q = 19
PRESET_WIND_TUNNEL = 0
PRESET_CAVITY = 1
SCHEME_AB = 0
SCHEME_AA = 1
SCHEME_SLIDE = 2
LAYOUT_DEFAULT = 0
LAYOUT_PLANES = 1
LAYOUT_ROWS = 2
TWO_STEP_AUTO = 0
TWO_STEP_OFF = 1

ABI_VERSION = 3   # include/lbm_abi.h LBM_ABI_VERSION

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liblbm_hip.so")


class LbmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"lbm error {code}: {msg}")
        self.code = code


class Opts(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("preset", C.c_int32),
                ("scheme", C.c_int32), ("z0", C.c_int32), ("depth_global", C.c_int32),
                ("stream", C.c_void_p), ("plane_skew_bytes", C.c_int32),
                ("layout", C.c_int32), ("boundary_stream", C.c_void_p),
                ("slide_chunk", C.c_int32), ("two_step", C.c_int32), ("reserved", C.c_int32 * 2)]


class Stats(C.Structure):
    _fields_ = [("steps", C.c_uint64), ("cells", C.c_uint64), ("bytes_per_cell", C.c_uint64),
                ("population_bytes", C.c_uint64), ("last_timed_ms", C.c_double),
                ("last_timed_steps", C.c_uint64), ("last_timed_launches", C.c_uint64),
                ("fused_cells", C.c_uint64), ("rim_cells", C.c_uint64),
                ("last_timed_fused_launches", C.c_uint64), ("last_timed_fused_ms", C.c_double),
                ("fused_wide_loads", C.c_int32), ("fused_tuned_ms", C.c_float * 2), ("fused_tail_cells", C.c_uint64)]


# every symbol include/lbm_abi.h declares, with its ctypes signature
_fp = C.POINTER(C.c_float)
_up = C.POINTER(C.c_uint8)
_ctx = C.c_void_p
SIGNATURES = {
    "lbm_create": ([C.POINTER(_ctx), C.c_int, C.c_int, C.c_int, C.POINTER(Opts)], C.c_int),
    "lbm_destroy": ([_ctx], C.c_int),
    "lbm_init": ([_ctx], C.c_int),
    "lbm_set_params": ([_ctx, C.c_float, C.c_float, C.c_float], C.c_int),
    "lbm_get_params": ([_ctx, _fp, _fp, _fp], C.c_int),
    "lbm_add_solid_rect": ([_ctx] + [C.c_int] * 6, C.c_int),
    "lbm_set_solid_u8": ([_ctx, _up, C.c_size_t], C.c_int),
    "lbm_set_solid_u8_device": ([_ctx, C.c_void_p, C.c_size_t], C.c_int),
    "lbm_device_copy": ([_ctx, C.c_void_p, C.c_void_p, C.c_size_t], C.c_int),
    "lbm_copy_to_host": ([_ctx, C.c_void_p, C.c_void_p, C.c_size_t], C.c_int),
    "lbm_write_field": ([_ctx, C.c_int, C.c_void_p, C.c_size_t, C.c_int], C.c_int),
    "lbm_load_mesh": ([_ctx, C.c_char_p], C.c_int),
    "lbm_step": ([_ctx, C.c_int], C.c_int),
    "lbm_compute_curl": ([_ctx], C.c_int),
    "lbm_step_frames": ([_ctx, C.c_int, C.c_int], C.c_int),
    "lbm_sync": ([_ctx], C.c_int),
    "lbm_set_streams": ([_ctx, C.c_void_p, C.c_void_p], C.c_int),
    "lbm_set_field_window": ([_ctx, C.c_int, C.c_int], C.c_int),
    "lbm_read_density": ([_ctx, _fp, C.c_size_t], C.c_int),
    "lbm_read_velocity": ([_ctx, _fp, C.c_size_t], C.c_int),
    "lbm_read_solid_u8": ([_ctx, _up, C.c_size_t], C.c_int),
    "lbm_read_velocity_strided": ([_ctx, C.c_int, C.c_int, C.c_int, _fp, C.c_size_t], C.c_int),
    "lbm_read_velocity_strided_noalloc": ([_ctx, C.c_int, C.c_int, C.c_int, _fp, C.c_size_t], C.c_int),
    "lbm_read_speed": ([_ctx, _fp, C.c_size_t], C.c_int),
    "lbm_read_curl": ([_ctx, _fp, C.c_size_t], C.c_int),
    "lbm_read_populations": ([_ctx, _fp, C.c_size_t], C.c_int),
    "lbm_write_populations": ([_ctx, _fp, C.c_size_t], C.c_int),
    "lbm_read_classification": ([_ctx, _up, C.c_size_t], C.c_int),
    "lbm_device_fields": ([_ctx, C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_up)], C.c_int),
    "lbm_get_dims": ([_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
    "lbm_get_stats": ([_ctx, C.POINTER(Stats)], C.c_int),
    "lbm_timer_begin": ([_ctx], C.c_int),
    "lbm_timer_end": ([_ctx], C.c_int),
    "lbm_last_error": ([_ctx], C.c_char_p),
    "lbm_abi_version": ([], C.c_int),
    "lbm_host_voxelize_obj": ([C.c_char_p, C.c_int, C.c_int, C.c_int, _up, C.c_size_t, _fp], C.c_int),
    "lbm_host_collide_cell": ([_fp, C.c_float, C.c_float, _fp, _fp], C.c_int),
    "lbm_halo_message_floats": ([_ctx, C.POINTER(C.c_size_t)], C.c_int),
    "lbm_halo_pack": ([_ctx, C.c_int, C.c_void_p], C.c_int),
    "lbm_halo_unpack": ([_ctx, C.c_int, C.c_void_p], C.c_int),
    "lbm_halo_get_mask_layer": ([_ctx, C.c_int, _up, C.c_size_t], C.c_int),
    "lbm_halo_set_mask_ghost": ([_ctx, C.c_int, _up, C.c_size_t], C.c_int),
    "lbm_halo_unpack_both": ([_ctx, C.c_void_p, C.c_void_p], C.c_int),
    "lbm_step_boundary_layers": ([_ctx, C.c_void_p, C.c_void_p], C.c_int),
    "lbm_step_interior_layers": ([_ctx], C.c_int),
    "lbm_step_finish": ([_ctx], C.c_int),
    "lbm_pair_available": ([_ctx, C.POINTER(C.c_int)], C.c_int),
    "lbm_pair_boundary_first": ([_ctx, C.c_void_p, C.c_void_p], C.c_int),
    "lbm_pair_interior": ([_ctx], C.c_int),
    "lbm_pair_boundary_second": ([_ctx, C.c_void_p, C.c_void_p], C.c_int),
    "lbm_pair_finish": ([_ctx], C.c_int),
    "lbm_fused_chain_stamps": ([_ctx, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_int)], C.c_int),
    "lbm_ipc_alloc": ([C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.c_void_p], C.c_int),
    "lbm_ipc_free": ([C.c_void_p], C.c_int),
    "lbm_ipc_open": ([C.c_void_p, C.POINTER(C.c_void_p)], C.c_int),
    "lbm_ipc_close": ([C.c_void_p], C.c_int),
    "lbm_ipc_copy": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t], C.c_int),
    "lbm_flag_sync": ([C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_int,
                       C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_int, C.c_uint32, C.c_void_p], C.c_int),
    "lbm_flag_peek": ([C.c_void_p, C.POINTER(C.c_uint32)], C.c_int),
    "lbm_flag_post": ([C.c_void_p, C.c_void_p, C.c_uint32], C.c_int),
    "lbm_flag_await": ([C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p], C.c_int),
}

_lib = None


def load_library(path=None):
    """dlopen the HIP library; raises if it is missing (no fallback)."""
    global _lib
    if _lib is None or path is not None:
        p = path or os.environ.get("LBM_LIB") or LIB_PATH   # LBM_LIB: an experimental build (tuning probes)
        if not os.path.exists(p):
            raise FileNotFoundError(
                f"{p} is missing: build it with `make -C cfd_amd/csrc` (or __graft_entry__.build()); "
                "the hot path has no CPU fallback")
        # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  Two HIP runtimes in one
        # process do not work ("No HIP GPUs are available"), so when torch is installed let it
        # load its runtime FIRST; our NEEDED libamdhip64.so.7 then binds to that same copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(p)
        for name, (args, res) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = res
        if lib.lbm_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{p}: ABI version {lib.lbm_abi_version()}, this binding is written for {ABI_VERSION} "
                               "(rebuild with `make -C cfd_amd/csrc`)")
        _lib = lib
    return _lib


def _f(a):
    return a.ctypes.data_as(_fp)


def _u(a):
    return a.ctypes.data_as(_up)


class Solver:
    """Mirror of `lbm_solver` (lbm_solver.cuh:17-51): ctor(W,H,D), init, step, reset,
    compute_curl, add_solid, load_mesh, density / velocity / solid-mask read-back, settings."""

    def __init__(self, W, H, D, preset=PRESET_WIND_TUNNEL, device=-1, z0=0, depth_global=0,
                 stream=None, plane_skew_bytes=-1, scheme=SCHEME_AB, boundary_stream=None,
                 layout=LAYOUT_DEFAULT, slide_chunk=0, two_step=TWO_STEP_AUTO):
        self._lib = load_library()
        self.W, self.H, self.D = W, H, D
        self.n = W * H * D
        self.nf = self.n          # cells inside the field window
        self.preset = preset
        o = Opts()
        o.struct_size = C.sizeof(Opts)
        o.device = device
        o.preset = preset
        o.scheme = scheme
        o.z0 = z0
        o.depth_global = depth_global
        o.stream = stream
        o.plane_skew_bytes = plane_skew_bytes
        o.layout = layout
        o.boundary_stream = boundary_stream
        o.slide_chunk = slide_chunk
        o.two_step = two_step
        self._ctx = _ctx()
        rc = self._lib.lbm_create(C.byref(self._ctx), W, H, D, C.byref(o))
        if rc != 0:
            msg = self._lib.lbm_last_error(None)
            self._ctx = None
            raise LbmError(rc, msg.decode() if msg else "?")

    def _chk(self, rc):
        if rc != 0:
            msg = self._lib.lbm_last_error(self._ctx)
            raise LbmError(rc, msg.decode() if msg else "?")

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.lbm_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference surface ---------------------------------------------------------------
    def init(self):
        self._chk(self._lib.lbm_init(self._ctx))

    reset = init  # lbm_solver.cu:496

    def set_params(self, inlet_velocity=None, omega_shear=None, omega_bulk=None):
        u, ws, wb = self.params()
        self._chk(self._lib.lbm_set_params(
            self._ctx,
            u if inlet_velocity is None else inlet_velocity,
            ws if omega_shear is None else omega_shear,
            wb if omega_bulk is None else omega_bulk))

    def params(self):
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        self._chk(self._lib.lbm_get_params(self._ctx, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def step(self, nsteps=1):
        self._chk(self._lib.lbm_step(self._ctx, nsteps))

    def step_frames(self, nframes=1, curl=False):
        """the reference's frame loop on the device: every step writes rho / u (and curl) as lbm_solver::step does"""
        self._chk(self._lib.lbm_step_frames(self._ctx, nframes, 1 if curl else 0))

    def compute_curl(self):
        self._chk(self._lib.lbm_compute_curl(self._ctx))

    def add_solid(self, x, y, z, w, h, d):
        self._chk(self._lib.lbm_add_solid_rect(self._ctx, x, y, z, w, h, d))

    def set_solid(self, mask_u8):
        m = np.ascontiguousarray(mask_u8, np.uint8).reshape(-1)
        self._chk(self._lib.lbm_set_solid_u8(self._ctx, _u(m), m.size))

    def load_mesh(self, path):
        self._chk(self._lib.lbm_load_mesh(self._ctx, os.fsencode(path)))

    def sync(self):
        self._chk(self._lib.lbm_sync(self._ctx))

    def set_field_window(self, z_begin, nz):
        self._chk(self._lib.lbm_set_field_window(self._ctx, z_begin, nz))
        self.nf = self.W * self.H * nz

    def density(self):
        out = np.empty(self.nf, np.float32)
        self._chk(self._lib.lbm_read_density(self._ctx, _f(out), out.size))
        return out

    def velocity(self):
        out = np.empty((self.nf, 4), np.float32)
        self._chk(self._lib.lbm_read_velocity(self._ctx, _f(out), self.nf))
        return out

    def velocity_strided(self, sx, sy, sz, noalloc=False):
        nz = self.nf // (self.W * self.H)
        shape = (-(-nz // max(sz, 1)), -(-self.H // max(sy, 1)), -(-self.W // max(sx, 1)))  # bad strides: the library rejects them
        out = np.empty(shape + (4,), np.float32)
        fn = self._lib.lbm_read_velocity_strided_noalloc if noalloc else self._lib.lbm_read_velocity_strided
        self._chk(fn(self._ctx, sx, sy, sz, _f(out), out.size // 4))
        return out

    def speed(self):
        out = np.empty(self.nf, np.float32)
        self._chk(self._lib.lbm_read_speed(self._ctx, _f(out), out.size))
        return out

    def solid_mask(self):
        out = np.empty(self.n, np.uint8)
        self._chk(self._lib.lbm_read_solid_u8(self._ctx, _u(out), out.size))
        return out

    def curl_field(self):
        out = np.empty(self.n, np.float32)
        self._chk(self._lib.lbm_read_curl(self._ctx, _f(out), out.size))
        return out

    def populations(self):
        out = np.empty((Q, self.n), np.float32)
        self._chk(self._lib.lbm_read_populations(self._ctx, _f(out), out.size))
        return out

    def set_populations(self, f):
        a = np.ascontiguousarray(f, np.float32).reshape(-1)
        self._chk(self._lib.lbm_write_populations(self._ctx, _f(a), a.size))

    def classification(self):
        cls = np.empty((self.n, Q), np.uint8)
        self._chk(self._lib.lbm_read_classification(self._ctx, _u(cls), cls.size))
        return cls

    # -- measurement ---------------------------------------------------------------------
    def stats(self):
        s = Stats()
        self._chk(self._lib.lbm_get_stats(self._ctx, C.byref(s)))
        out = {k: getattr(s, k) for k, _ in Stats._fields_}
        out["fused_tuned_ms"] = [float(x) for x in s.fused_tuned_ms]
        return out

    def fused_chain_stamps(self):
        """(start, end) device-clock stamps (100 MHz) of every workgroup of one k_fused2 launch, shape (n, 2);
        None when the two-steps-per-pass path is not in use."""
        n = C.c_int(0)
        self._chk(self._lib.lbm_fused_chain_stamps(self._ctx, None, 0, C.byref(n)))
        if n.value == 0:
            return None
        out = np.zeros((n.value, 2), np.uint64)
        self._chk(self._lib.lbm_fused_chain_stamps(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size, C.byref(n)))
        return out

    def timer_begin(self):
        self._chk(self._lib.lbm_timer_begin(self._ctx))

    def timer_end(self):
        self._chk(self._lib.lbm_timer_end(self._ctx))
        return self.stats()["last_timed_ms"]


def host_voxelize_obj(path, W, H, D):
    """Product voxeliser (host stage of load_mesh); returns (mask u8[n], scale, rc)."""
    L = load_library()
    mask = np.zeros(W * H * D, np.uint8)
    sc = C.c_float(0)
    rc = L.lbm_host_voxelize_obj(os.fsencode(path), W, H, D, _u(mask), mask.size, C.byref(sc))
    return mask, sc.value, rc


def host_collide_cell(f19, omega_shear, omega_bulk):
    L = load_library()
    f = np.ascontiguousarray(f19, np.float32)
    out = np.zeros(19, np.float32)
    ru = np.zeros(4, np.float32)
    rc = L.lbm_host_collide_cell(_f(f), omega_shear, omega_bulk, _f(out), _f(ru))
    assert rc == 0
    return out, ru
