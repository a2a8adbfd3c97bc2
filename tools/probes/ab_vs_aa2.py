import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cfd_amd import lbm, slab
dims = (1024, 512, 512)
W, H, D = dims
engines = [slab.HipSlabEngine(W, H, D, z0, 128, 0) for z0 in (0, 128, 256, 384)]
cl = slab.LocalSlabCluster(engines, 0)
for e in engines:
    e.solver.init()
cl.exchange_mask(); cl.exchange_populations(); cl.step(5)
ref = np.concatenate([e.solver.density() for e in engines])
for e in engines:
    e.solver.close()
for scheme in (0, 1):
    s = lbm.Solver(*dims, scheme=scheme)
    s.init(); s.step(5)
    r = s.density(); s.close()
    bad = np.nonzero(r != ref)[0]
    print("scheme", scheme, "mismatches vs 4-slab A/B:", bad.size, flush=True)
    if bad.size:
        x = bad % W; y = (bad // W) % H; z = bad // (W * H)
        print("  z histogram (nonzero):", {int(k): int(v) for k, v in zip(*np.unique(z, return_counts=True))} if np.unique(z).size < 40 else (int(z.min()), int(z.max()), np.unique(z).size))
        print("  x range", int(x.min()), int(x.max()), "y range", int(y.min()), int(y.max()))
        print("  sample", bad[:5], r[bad[:5]], ref[bad[:5]])
