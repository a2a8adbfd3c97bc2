"""Is the Python slab driver host-bound?  Time to ENQUEUE n steps of a one-rank ring (transport that moves nothing) against
the time until the GPU has finished them.  usage: [W H D] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfd_amd import lbm, slab
W, H, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 256, 256)
n = int(sys.argv[4]) if len(sys.argv) > 4 else 201
eng = slab.HipSlabEngine(W, H, D, 0, D, lbm.PRESET_WIND_TUNNEL, two_streams=True)
eng.solver.init()
transport = slab.NoExchange()
if os.environ.get("PROBE_RCCL") == "1":          # the ring of one rank to itself over RCCL, as bench.py --rehearse-slab
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    transport = slab.TorchP2P()
elif os.environ.get("PROBE_RCCL") == "copy":     # device copies on the side stream
    transport = slab.LoopbackP2P()
drv = slab.SlabSolver(eng, lbm.PRESET_WIND_TUNNEL, 1, 0, transport)
if os.environ.get("PROBE_AUTOTUNE") == "1":
    print("autotune:", [(i, j, round(t * 1e6, 1)) for i, j, t in drv.autotune_streams()], flush=True)
drv.exchange_mask(); drv.exchange_populations()
drv.step(101); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); drv.step(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{W}x{H}x{D}: {n} steps enqueued in {(t1 - t0) / n * 1e6:.1f} us per step, finished in {(t2 - t0) / n * 1e6:.1f} us per step", flush=True)
eng.solver.close()
