import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cfd_amd import lbm
for dims in ((1024, 64, 64), (1024, 512, 16), (1024, 16, 512), (512, 512, 512), (1024, 512, 512), (1024, 512, 128), (1024, 256, 256)):
    r = []
    for scheme in (0, 1):
        s = lbm.Solver(*dims, scheme=scheme)
        s.init(); s.step(5)
        r.append(s.density()); s.close()
    bad = np.nonzero(r[0] != r[1])[0]
    msg = ""
    if bad.size:
        W, H, D = dims
        x = bad % W; y = (bad // W) % H; z = bad // (W * H)
        msg = f" x[{x.min()},{x.max()}] y[{y.min()},{y.max()}] z[{z.min()},{z.max()}]"
    print(dims, "mismatches:", bad.size, msg, flush=True)
