"""Long-run agreement: the in-place AA kernel, a 4-slab A/B run, the sliding scheme and a 3-slab sliding run
against the plain A/B run after many steps
(bit for bit), on BASELINE.json configs[2] (512x256x256 wind tunnel + car stand-in).  One-off probe."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cfd_amd import lbm, slab
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
W, H, D = 512, 256, 256
car = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "car_like.obj")
t0 = time.time()


def same(x, y):  # bit patterns, so that NaNs of a diverged run compare too
    return np.array_equal(x.view(np.uint32), y.view(np.uint32))


a = lbm.Solver(W, H, D); a.init(); a.load_mesh(car); a.step(steps)
rho, u = a.density(), a.velocity(); a.close()
print(f"A/B  {steps} steps: sum rho {rho.astype(np.float64).sum():.6f}, max|u| {np.abs(u).max():.6f}, finite {np.isfinite(rho).all() and np.isfinite(u).all()}  ({time.time()-t0:.0f} s)", flush=True)
b = lbm.Solver(W, H, D, scheme=lbm.SCHEME_AA); b.init(); b.load_mesh(car); b.step(steps)
print("AA   identical to A/B:", same(b.density(), rho) and same(b.velocity(), u), flush=True)
b.close()
engines = [slab.HipSlabEngine(W, H, D, z0, 64, 0) for z0 in (0, 64, 128, 192)]
cl = slab.LocalSlabCluster(engines, 0)
for e in engines:
    e.solver.init(); e.solver.load_mesh(car)
cl.exchange_mask(); cl.exchange_populations(); cl.step(steps)
print("4 slabs identical to A/B:", same(np.concatenate([e.solver.density() for e in engines]), rho)
      and same(np.concatenate([e.solver.velocity() for e in engines]), u), f"({time.time()-t0:.0f} s)", flush=True)
for e in engines:
    e.solver.close()
c = lbm.Solver(W, H, D, scheme=lbm.SCHEME_SLIDE); c.init(); c.load_mesh(car); c.step(steps)
print("sliding identical to A/B:", same(c.density(), rho) and same(c.velocity(), u), flush=True)
c.close()
engines = [slab.HipSlabEngine(W, H, D, z0, d, 0, scheme=lbm.SCHEME_SLIDE) for z0, d in ((0, 100), (100, 56), (156, 100))]
cl = slab.LocalSlabCluster(engines, 0)
for e in engines:
    e.solver.init(); e.solver.load_mesh(car)
cl.exchange_mask(); cl.exchange_populations(); cl.step(steps)
print("3 sliding slabs identical to A/B:", same(np.concatenate([e.solver.density() for e in engines]), rho)
      and same(np.concatenate([e.solver.velocity() for e in engines]), u), f"({time.time()-t0:.0f} s)", flush=True)
