"""What does cutting a domain into z-slabs cost in kernel time?  All slabs of config 3 (1024x512x512 + car)
run on ONE GPU one after the other (device copies as transport): time per step of the whole domain against
the undecomposed run.  (ad-hoc probe; the real exchange runs over RCCL between GPUs)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cfd_amd import lbm, slab
W, H, Dg = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 512, 512)
car = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "car_like.obj")
for P in (1, 2, 4, 8):
    D = Dg // P
    engines = [slab.HipSlabEngine(W, H, Dg, r * D, D, lbm.PRESET_WIND_TUNNEL) for r in range(P)]
    cl = slab.LocalSlabCluster(engines, lbm.PRESET_WIND_TUNNEL)
    for e in engines:
        e.solver.init(); e.solver.load_mesh(car)
    cl.exchange_mask(); cl.exchange_populations()
    cl.step(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); cl.step(10); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{W}x{H}x{Dg} in {P} slabs of {D} layers on one GPU: {dt*1e3:.3f} ms/step = {W*H*Dg/dt/1e6:.0f} MLUPS (per slab {dt/P*1e3:.3f} ms)", flush=True)
    for e in engines:
        e.solver.close()
