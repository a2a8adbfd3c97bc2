"""Cost of the boundary (general) path: the same domain with every tile forced onto it, and the share
of tiles that take it in the car configuration (ad-hoc probe)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cfd_amd import lbm
car = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "car_like.obj")
for dims, preset, mesh in (((256, 256, 256), 1, None), ((256, 256, 256), 0, None), ((512, 256, 256), 0, car)):
    for force in (0, 1):
        if force: os.environ["LBM_FORCE_GENERAL"] = "1"
        else: os.environ.pop("LBM_FORCE_GENERAL", None)
        s = lbm.Solver(*dims, preset=preset)
        s.set_field_window(0, 0)
        s.init()
        if mesh: s.load_mesh(mesh)
        s.step(6); s.sync()
        s.timer_begin(); s.step(20); ms = s.timer_end() / 20
        m = s.solid_mask().reshape(dims[2], dims[1], dims[0])
        fluid = int((m == 0).sum())
        tiles = m.reshape(dims[2], dims[1], dims[0] // 64, 64)
        nsol = (tiles > 0).sum(axis=3)
        print(f"{dims} preset={preset} {'car' if mesh else 'empty'} force_general={force}: {ms*1e3:.1f} us, "
              f"{fluid*152/ms/1e9:.3f} TB/s (fluid cells); tiles all-solid {float((nsol==64).mean()):.3f}, partly solid {float(((nsol>0)&(nsol<64)).mean()):.3f}", flush=True)
        s.close()
