import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.distributed as dist
from cfd_amd import lbm, slab
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
a = torch.arange(1000, dtype=torch.float32, device="cuda"); b = a + 5000
ra = torch.zeros_like(a); rb = torch.zeros_like(a)
ops = [dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.isend, b, 0), dist.P2POp(dist.irecv, ra, 0), dist.P2POp(dist.irecv, rb, 0)]
for w in dist.batch_isend_irecv(ops): w.wait()
torch.cuda.synchronize()
print("self p2p order: first recv got first send:", bool((ra == a).all()), " second:", bool((rb == b).all()), "| swapped:", bool((ra == b).all()))
stream = torch.cuda.current_stream().cuda_stream
W, H, D = 64, 48, 40
for preset in (0, 1):
    ref = lbm.Solver(W, H, D, preset=preset); ref.init(); ref.add_solid(20, 0, 10, 8, 20, 20)
    eng = slab.HipSlabEngine(W, H, D, 0, D, preset, device=0)
    drv = slab.SlabSolver(eng, preset, 1, 0, slab.TorchP2P())
    eng.solver.init(); eng.solver.add_solid(20, 0, 10, 8, 20, 20)
    drv.exchange_mask(); drv.exchange_populations()
    for i in range(6):
        ref.step(1); drv.step(1); torch.cuda.synchronize()
        ok = np.array_equal(ref.density(), eng.solver.density())
        bad = np.nonzero(ref.density() != eng.solver.density())[0]
        print(preset, "step", i, ok, len(bad), [(int(j % W), int(j // W % H), int(j // (W * H))) for j in bad[:6]])
dist.destroy_process_group()
