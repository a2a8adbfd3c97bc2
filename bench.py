#!/usr/bin/env python3
"""bench.py -- headline benchmark of the D3Q19 hot path (BASELINE.json: MLUPS + % of HBM roofline).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one fused stream-collide update of the whole domain (lbm_step(ctx, 1), 152 algorithmic
bytes per cell: 19 fp32 populations read + written; rho/u are not written per step).
N = 1: BASELINE.json configs[1], the 256^3 lid-driven cavity, populations resident in HBM.
N > 1: weak scaling -- every rank owns a 256^3 z-slab of a 256 x 256 x (256 N) cavity and exchanges
the five face-crossing populations per face per step with its neighbours by RCCL send/recv
(torch.distributed backend "nccl"), overlapped with the interior update.
Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events on the solver's stream;
`cpu_baseline` times the CPU oracle (a port of the reference update: the reference has no CPU path)
on a bounded sample of the same workload, on rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_CELL = 152           # 19 populations x 4 B x (read + write)


def cpu_baseline(W, H, D, preset, budget_s=12.0):
    """Oracle (kind "port") on the host cores: same 256^3 cavity state, as many steps as fit the
    budget (at least 2)."""
    from oracle import oracle as O
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    threads = max(1, min(threads, 16))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    O.build()
    s = O.Solver(W, H, D, preset)
    s.init()
    s.step(1)                                   # touch every page
    t0 = time.perf_counter()
    steps = 0
    while steps < 2 or (time.perf_counter() - t0 < budget_s and steps < 50):
        s.step(1)
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": round(W * H * D * steps / dt / 1e6, 2), "unit": "MLUPS", "cores": threads,
            "kind": "port",
            "sample": f"{W}x{H}x{D} cavity, {steps} steps of oracle/lbm_oracle.c (OpenMP over z, "
                      f"{threads} threads, -O2 -ffp-contract=off), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, nargs=3, default=[256, 256, 256], help="per-GPU W H D")
    ap.add_argument("--preset", choices=["cavity", "wind_tunnel"], default="cavity")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from cfd_amd import lbm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    W, H, D = args.size
    preset = lbm.PRESET_CAVITY if args.preset == "cavity" else lbm.PRESET_WIND_TUNNEL

    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        from cfd_amd import slab
        eng = slab.HipSlabEngine(W, H, D * world, rank * D, D, preset, device=local_rank)
        solver = eng.solver
        drv = slab.SlabSolver(eng, preset, world, rank, slab.TorchP2P())
        solver.init()
        drv.exchange_mask()
        drv.exchange_populations()
        step = drv.step
    else:
        solver = lbm.Solver(W, H, D, preset=preset, device=local_rank)
        solver.init()
        step = solver.step

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step(args.warmup)
    fence()
    solver.timer_begin()
    t0 = time.perf_counter()
    step(args.steps)
    dev_ms = solver.timer_end()                 # HIP events on the solver's stream
    fence()
    dt = time.perf_counter() - t0

    if dist is not None:
        tmax = torch.tensor([dt, dev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt, dev_ms = float(tmax[0]), float(tmax[1])

    cells = W * H * D
    mlups_total = cells * world * args.steps / dt / 1e6
    ms_per_step = dt / args.steps * 1e3
    # dominant kernel: k_stream_collide; one launch per step at N=1 (three at N>1: 2 boundary
    # layers + interior); achieved = algorithmic bytes per step / device time per step
    dev_ms_per_step = dev_ms / args.steps
    achieved = cells * BYTES_PER_CELL / (dev_ms_per_step * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f"{W}x{H}x{D}_{args.preset}")
        except Exception:
            traffic = None

    if rank == 0:
        out = {
            "metric": "MLUPS (D3Q19 fused stream-collide, whole job)",
            "value": round(mlups_total, 1), "unit": "MLUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.preset} {W}x{H}x{D * world} D3Q19 fp32, A/B pull, "
                                   f"{'z-slabs ' + str(world) + 'x' + str(D) + ' layers, RCCL halo' if world > 1 else 'one GPU'}"
                                   " (BASELINE.json configs[1])",
                       "per_gpu_cells": cells, "bytes_per_cell": BYTES_PER_CELL,
                       "moments_per_step": False},
            "mlups_per_gpu": round(mlups_total / world, 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "kernel": "k_stream_collide",
                         "launch_ms": round(dev_ms_per_step, 5),
                         "algorithmic_bytes_per_launch": cells * BYTES_PER_CELL},
        }
        if world == 1 and not args.no_cpu_baseline:
            solver.close()
            out["cpu_baseline"] = cpu_baseline(W, H, D, preset)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
