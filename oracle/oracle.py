"""ctypes loader for the CPU oracle (oracle/lbm_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing
under cfd_amd/ does.  Every function mirrors one reference routine (file:line in lbm_oracle.c).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liblbm_oracle.so")

Q = 19
PRESET_WIND_TUNNEL = 0
PRESET_CAVITY = 1


def build(force=False):
    src = os.path.join(_HERE, "lbm_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "CC=gcc"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        fp = C.POINTER(C.c_float)
        up = C.POINTER(C.c_ubyte)
        ip = C.POINTER(C.c_int)
        L.orc_tables.argtypes = [fp, ip, ip, ip, ip, ip, ip]
        L.orc_init.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int]
        L.orc_step.argtypes = [fp, fp, fp, fp, up, C.c_int, C.c_int, C.c_int,
                               C.c_float, C.c_float, C.c_float, C.c_int]
        L.orc_run.argtypes = [fp, fp, fp, fp, up, C.c_int, C.c_int, C.c_int,
                              C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]
        L.orc_classify.argtypes = [up, C.c_int, C.c_int, C.c_int, C.c_int, up, up]
        L.orc_curl.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int]
        L.orc_add_solid_rect.argtypes = [up] + [C.c_int] * 9
        L.orc_load_mesh.argtypes = [C.c_char_p, up, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.orc_load_mesh.restype = C.c_int
        L.orc_step_slab.argtypes = [fp, fp, fp, fp, up] + [C.c_int] * 5 + [C.c_float] * 3 + [C.c_int] * 3
        L.orc_collide_cell.argtypes = [fp, C.c_float, C.c_float, fp, fp]
        L.orc_fnv1a64.argtypes = [up, C.c_size_t]
        L.orc_fnv1a64.restype = C.c_ulonglong
        L.orc_mesh_new.restype = C.c_void_p
        L.orc_mesh_free.argtypes = [C.c_void_p]
        L.orc_mesh_load_obj.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_mesh_load_obj.restype = C.c_int
        L.orc_mesh_ntris.argtypes = [C.c_void_p]
        L.orc_mesh_ntris.restype = C.c_size_t
        L.orc_mesh_get_tris.argtypes = [C.c_void_p, fp]
        L.orc_mesh_bounds.argtypes = [C.c_void_p, fp]
        L.orc_mesh_rotate_y.argtypes = [C.c_void_p, C.c_float]
        L.orc_mesh_voxelize.argtypes = [C.c_void_p, up, C.c_int, C.c_int, C.c_int, C.c_float]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_ubyte))


def tables():
    w = np.zeros(Q, np.float32)
    t = [np.zeros(Q, np.int32) for _ in range(6)]
    ip = C.POINTER(C.c_int)
    lib().orc_tables(_fp(w), *[a.ctypes.data_as(ip) for a in t])
    return dict(w=w, cx=t[0], cy=t[1], cz=t[2], inv=t[3], rfy=t[4], rfz=t[5])


class Solver:
    """Host mirror of the reference `lbm_solver` (lbm_solver.cuh:17-51) on the CPU oracle.

    Same call surface as cfd_amd.lbm.Solver so parity tests drive both with identical code.
    """

    def __init__(self, W, H, D, preset=PRESET_WIND_TUNNEL):
        self.W, self.H, self.D = W, H, D
        self.n = W * H * D
        self.preset = preset
        self.inlet_velocity = np.float32(0.08)   # lbm_solver.cuh:11-13
        self.omega_shear = np.float32(1.98)
        self.omega_bulk = np.float32(1.90)
        self.f = np.zeros(Q * self.n, np.float32)
        self.f_new = np.zeros(Q * self.n, np.float32)
        self.rho = np.zeros(self.n, np.float32)
        self.u = np.zeros(4 * self.n, np.float32)
        self.curl = np.zeros(self.n, np.float32)
        self.solid = np.zeros(self.n, np.uint8)

    def init(self):
        lib().orc_init(_fp(self.f), _fp(self.rho), _fp(self.u), self.W, self.H, self.D)

    reset = init

    def set_params(self, inlet_velocity=None, omega_shear=None, omega_bulk=None):
        if inlet_velocity is not None:
            self.inlet_velocity = np.float32(inlet_velocity)
        if omega_shear is not None:
            self.omega_shear = np.float32(omega_shear)
        if omega_bulk is not None:
            self.omega_bulk = np.float32(omega_bulk)

    def step(self, nsteps=1):
        for _ in range(nsteps):
            lib().orc_step(_fp(self.f), _fp(self.f_new), _fp(self.rho), _fp(self.u),
                           _up(self.solid), self.W, self.H, self.D,
                           self.inlet_velocity, self.omega_shear, self.omega_bulk, self.preset)
            self.f, self.f_new = self.f_new, self.f   # lbm_solver.cu:490

    def compute_curl(self):
        lib().orc_curl(_fp(self.u), _fp(self.curl), self.W, self.H, self.D)

    def add_solid(self, x, y, z, w, h, d):
        lib().orc_add_solid_rect(_up(self.solid), self.W, self.H, self.D, x, y, z, w, h, d)

    def set_solid(self, mask_u8):
        self.solid[:] = np.asarray(mask_u8, np.uint8).reshape(-1)

    def load_mesh(self, path, floor_plane=True):
        sc = C.c_float(0)
        rc = lib().orc_load_mesh(os.fsencode(path), _up(self.solid), self.W, self.H, self.D,
                                 int(bool(floor_plane)), C.byref(sc))
        self.mesh_scale = sc.value
        return rc

    def density(self):
        return self.rho.copy()

    def velocity(self):
        return self.u.reshape(self.n, 4).copy()

    def solid_mask(self):
        return self.solid.copy()

    def curl_field(self):
        return self.curl.copy()

    def populations(self):
        """Current populations in the reference layout f[k*n + idx]."""
        return self.f.reshape(Q, self.n).copy()

    def set_populations(self, f):
        """Import populations in the reference layout (what cfd_amd.lbm.Solver.set_populations takes)."""
        self.f[:] = np.asarray(f, np.float32).reshape(-1)

    def classification(self):
        cls = np.zeros((self.n, Q), np.uint8)
        sol = np.zeros(self.n, np.uint8)
        lib().orc_classify(_up(self.solid), self.W, self.H, self.D, self.preset, _up(cls), _up(sol))
        return cls, sol


def fnv1a64(buf):
    b = np.ascontiguousarray(buf, np.uint8)
    return int(lib().orc_fnv1a64(_up(b), b.size))


def load_obj_triangles(path):
    L = lib()
    m = L.orc_mesh_new()
    try:
        rc = L.orc_mesh_load_obj(m, os.fsencode(path))
        nt = L.orc_mesh_ntris(m)
        tris = np.zeros((nt, 9), np.float32)
        if nt:
            L.orc_mesh_get_tris(m, _fp(tris))
        b = np.zeros(6, np.float32)
        L.orc_mesh_bounds(m, _fp(b))
        return rc, tris, b
    finally:
        L.orc_mesh_free(m)


def collide_cell(f19, omega_shear, omega_bulk):
    f = np.ascontiguousarray(f19, np.float32)
    out = np.zeros(Q, np.float32)
    ru = np.zeros(4, np.float32)
    lib().orc_collide_cell(_fp(f), omega_shear, omega_bulk, _fp(out), _fp(ru))
    return out, ru


def mesh_triangles(path):
    """The triangles the OBJ reader produces (quad split / ear clipping applied), as an (n, 9) float32 array."""
    L = lib()
    m = L.orc_mesh_new()
    try:
        if L.orc_mesh_load_obj(m, os.fsencode(path)) != 0:
            raise FileNotFoundError(path)
        n = L.orc_mesh_ntris(m)
        out = np.zeros((n, 9), np.float32)
        if n:
            L.orc_mesh_get_tris(m, _fp(out))
        return out
    finally:
        L.orc_mesh_free(m)


def voxelize_obj(path, W, H, D, floor_plane=True):
    mask = np.zeros(W * H * D, np.uint8)
    sc = C.c_float(0)
    rc = lib().orc_load_mesh(os.fsencode(path), _up(mask), W, H, D, int(bool(floor_plane)), C.byref(sc))
    return mask, sc.value, rc
