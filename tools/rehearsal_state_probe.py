#!/usr/bin/env python3
"""Why some slabs of bench.py's 8-slab rehearsal report `state_finite: false`, and whether that touches their timing.

A rehearsed slab's RCCL ring closes on itself: its lower ghost layer is its own top layer.  Where the car's cross-section
differs between those two faces (the slabs through its flanks) that is an abrupt change of geometry every step, at the
reference's omega_shear = 1.98.  This probe steps such a slab with the reference's parameters and with tame ones
(same launches, same plan) and prints finiteness and time per step of both.

    python tools/rehearsal_state_probe.py [--slab 2] [--steps 20]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slab", type=int, nargs="*", default=[2, 3])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--min-seconds", type=float, default=0.3)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import bench
    from cfd_amd import lbm, slab
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(bench.free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    W, H, D = bench.CONFIGS[3][0]
    kw = dict(steps=args.steps, warmup=5, world=1, rank=0, local_rank=0, dist=dist, min_seconds=args.min_seconds,
              want_exposed=False)
    for r in args.slab:
        for name, params in (("reference defaults", None),
                             ("tame (U 0.02, omega 1.2 / 1.2)", dict(inlet_velocity=0.02, omega_shear=1.2, omega_bulk=1.2))):
            one = bench.run_slabs(lbm, (W, H, D), "wind_tunnel", "ab", "car", rehearse=(r, 8), params=params, **kw)
            print(json.dumps({"slab": r, "params": name, "steps": one["steps"], "ms_per_step": one["ms_per_step"],
                              "k_fused2_launch_ms": one["roofline"].get("launch_ms"), "state_finite": one["state_finite"]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
