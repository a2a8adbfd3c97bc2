import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfd_amd import lbm, slab
W, H, D = 512, 256, 256
car = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "car_like.obj")
a = lbm.Solver(W, H, D); a.init(); a.load_mesh(car); a.step(21)
rho, u = a.density(), a.velocity(); a.close()
engines = [slab.HipSlabEngine(W, H, D, z0, d, 0) for z0, d in ((0, 64), (64, 100), (164, 92))]
cl = slab.LocalSlabCluster(engines, 0)
for e in engines:
    e.solver.init(); e.solver.load_mesh(car)
cl.exchange_mask(); cl.exchange_populations()
paired = cl.step_pairs(10); cl.step(1)
print("paired", paired, [e.solver.stats()["fused_cells"] for e in engines])
r2 = np.concatenate([e.solver.density() for e in engines]); u2 = np.concatenate([e.solver.velocity() for e in engines])
print("3 slabs in pairs identical to the undecomposed run:", np.array_equal(r2.view(np.uint32), rho.view(np.uint32)) and np.array_equal(u2.view(np.uint32), u.view(np.uint32)))
