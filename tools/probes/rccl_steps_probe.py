"""Host enqueue vs total time of the RCCL slab step as the step count grows (one rank, ring to itself)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from cfd_amd import lbm, slab
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
W, H, D = 256, 256, 256
for two in (True, False, True, True):
    eng = slab.HipSlabEngine(W, H, D, 0, D, lbm.PRESET_WIND_TUNNEL, device=0, two_streams=two)
    drv = slab.SlabSolver(eng, lbm.PRESET_WIND_TUNNEL, 1, 0, slab.TorchP2P())
    eng.solver.init(); drv.exchange_mask(); drv.exchange_populations()
    drv.step(20); torch.cuda.synchronize()
    for n in (100, 200):
        if n == 'barrier':
            dist.barrier(); torch.cuda.synchronize(); print('-- after dist.barrier()'); continue
        if n == 'timer':
            eng.solver.timer_begin(); print('-- after timer_begin()'); continue
        t0 = time.perf_counter(); drv.step(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"two_streams={two} steps={n}: host enqueue {1e6*(t1-t0)/n:.0f} us/step, total {1e6*(t2-t0)/n:.0f} us/step", flush=True)
    eng.solver.close()
dist.destroy_process_group()
