"""The engine's frame loop without the window: one step, then rho/u on the device for the renderer
(lbm_device_fields), every frame.  A/B recomputes rho/u on demand (96 B/cell) unless it expects the
caller to ask again, in which case the step writes them as it goes (20 B/cell).  usage: W H D [scheme]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from cfd_amd import lbm
W, H, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (400, 100, 100)
scheme = int(sys.argv[4]) if len(sys.argv) > 4 else 0
s = lbm.Solver(W, H, D, scheme=scheme)
s.init(); s.load_mesh(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cube.obj"))
lib = s._lib
fp = C.POINTER(C.c_float)
rho, u, curl, solid = fp(), fp(), fp(), C.POINTER(C.c_uint8)()


def frame():
    s.step(1)
    s._chk(lib.lbm_device_fields(s._ctx, C.byref(rho), C.byref(u), None, None))


for _ in range(20):
    frame()
s.sync()
n = 500
t0 = time.perf_counter()
for _ in range(n):
    frame()
s.sync()
dt = (time.perf_counter() - t0) / n
print(f"{W}x{H}x{D} scheme={scheme}: {dt*1e6:.1f} us per frame (step + rho/u on the device), {W*H*D/dt/1e6:.0f} MLUPS", flush=True)
s.step(n); s.sync()
t0 = time.perf_counter(); s.step(n); s.sync()
dt = (time.perf_counter() - t0) / n
print(f"{W}x{H}x{D} scheme={scheme}: {dt*1e6:.1f} us per step without fields, {W*H*D/dt/1e6:.0f} MLUPS", flush=True)
