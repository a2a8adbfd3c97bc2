"""Single-GPU probe of the RCCL halo path: one rank, wind-tunnel ring => the rank exchanges with
itself through torch.distributed/nccl.  Checks bit-identity against the undecomposed solver and
reports host+device time per step of the overlapped schedule."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as dist
from cfd_amd import lbm, slab

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
stream = torch.cuda.current_stream().cuda_stream
W, H, D = 64, 48, 40
ref = lbm.Solver(W, H, D); ref.init(); ref.add_solid(20, 0, 10, 8, 20, 20); ref.step(12)
eng = slab.HipSlabEngine(W, H, D, 0, D, lbm.PRESET_WIND_TUNNEL, device=0)
drv = slab.SlabSolver(eng, lbm.PRESET_WIND_TUNNEL, 1, 0, slab.TorchP2P())
eng.solver.init(); eng.solver.add_solid(20, 0, 10, 8, 20, 20)
drv.exchange_mask(); drv.exchange_populations(); drv.step(12)
torch.cuda.synchronize()
print("self-ring bit-identical:", np.array_equal(ref.density(), eng.solver.density()) and np.array_equal(ref.velocity(), eng.solver.velocity()), flush=True)
eng = slab.HipSlabEngine(W, H, D, 0, D, lbm.PRESET_WIND_TUNNEL, device=0, scheme=lbm.SCHEME_AA)
drv = slab.SlabSolver(eng, lbm.PRESET_WIND_TUNNEL, 1, 0, slab.TorchP2P())
eng.solver.init(); eng.solver.add_solid(20, 0, 10, 8, 20, 20)
drv.exchange_mask(); drv.exchange_populations(); drv.step(12)
torch.cuda.synchronize()
print("AA self-ring bit-identical:", np.array_equal(ref.density(), eng.solver.density()) and np.array_equal(ref.velocity(), eng.solver.velocity()), flush=True)
for dims in [(256, 256, 256), (256, 256, 64), (1024, 512, 64)]:
    W, H, D = dims
    for preset in (lbm.PRESET_WIND_TUNNEL, lbm.PRESET_CAVITY):
        eng = slab.HipSlabEngine(W, H, D, 0, D, preset, device=0)
        drv = slab.SlabSolver(eng, preset, 1, 0, slab.TorchP2P())
        eng.solver.init(); drv.exchange_mask(); drv.exchange_populations()
        drv.step(20); torch.cuda.synchronize()
        t0 = time.perf_counter(); drv.step(100); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{dims} preset={preset}: host enqueue {1e6*(t1-t0)/100:.0f} us/step, total {1e6*(t2-t0)/100:.0f} us/step -> {W*H*D*100/(t2-t0)/1e6:.0f} MLUPS", flush=True)
        eng.solver.close()
dist.destroy_process_group()
