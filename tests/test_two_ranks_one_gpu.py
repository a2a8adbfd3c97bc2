"""SlabSolver across REAL processes on the HIP engine: two (three) ranks share cuda:0, rendezvous over gloo, halo
messages staged through the host (cfd_amd/slab.py HostStagedP2P; RCCL refuses two ranks on one device).  Everything
above the transport is what an N-GPU run executes: per-rank HipSlabEngine on two streams, the pair schedule next to
single split steps, ring neighbours, mask ghost exchange.  Must equal the undecomposed oracle bit for bit."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _rank(rank, world, port, preset, W, H, splits, steps, mask, params, out, transport="host"):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from cfd_amd import slab
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    Dg = sum(splits)
    z0, d = sum(splits[:rank]), splits[rank]
    e = slab.HipSlabEngine(W, H, Dg, z0, d, preset, device=0, two_streams=True)
    s = e.solver
    try:
        s.init()
        s.set_params(**params)
        s.set_solid(mask.reshape(Dg, H, W)[z0:z0 + d])
        t = slab.HostStagedP2P()
        if transport == "ipc":
            t = slab.IpcCopyP2P(t, dist, rank, world, timeout_ms=5000)
        drv = slab.SlabSolver(e, preset, world, rank, t)
        if transport == "ipc":
            assert t.links is not None, f"bind failed: {t.error}"
            assert t.verify(), f"verify failed: {t.error}"
        drv.exchange_mask()
        drv.exchange_populations()
        for n in steps:
            drv.step(n)
        torch.cuda.synchronize()
        lost = transport == "ipc" and t.timed_out()
        out[rank] = (z0, d, s.density(), s.velocity(), s.populations(), s.stats()["fused_cells"], lost)
        dist.barrier()
        if transport == "ipc":
            t.close()
    finally:
        s.close()
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("preset,splits", [(0, (16, 16)), (1, (20, 12)), (0, (12, 8, 12))])
def test_ranks_sharing_a_gpu_equal_the_oracle(oracle, preset, splits):
    import torch.multiprocessing as mp
    world = len(splits)
    W, H, Dg = 136, 44, sum(splits)
    steps = (5, 4, 1, 3)
    params = dict(inlet_velocity=0.06, omega_shear=1.7, omega_bulk=1.5)
    o = oracle.Solver(W, H, Dg, preset)
    o.init()
    o.add_solid(20, 0, 4, 9, 6, 25)            # crosses every interface
    o.add_solid(W - 2, 24, 9, 2, 5, 8)         # in the outlet column
    mask = o.solid_mask()
    o.set_params(**params)
    o.step(sum(steps))
    port = 29300 + (os.getpid() % 300) + 5 * world + preset
    out = mp.Manager().dict()
    mp.spawn(_rank, args=(world, port, preset, W, H, splits, steps, mask, params, out), nprocs=world, join=True)
    parts = [out[r] for r in range(world)]
    rho = np.concatenate([p[2] for p in parts])
    u = np.concatenate([p[3] for p in parts])
    f = np.concatenate([p[4].reshape(19, p[1], H * W) for p in parts], axis=1).reshape(19, -1)
    fluid = mask == 0
    np.testing.assert_array_equal(rho, o.density())
    np.testing.assert_array_equal(u, o.velocity())
    np.testing.assert_array_equal(f[:, fluid], o.populations()[:, fluid])
    assert any(p[5] > 0 for p in parts), "no rank took the pair schedule"


@pytest.mark.timeout(300)
@pytest.mark.parametrize("preset,splits", [(0, (16, 16)), (1, (12, 8, 12))])
def test_peer_copy_transport_between_processes(oracle, preset, splits):
    """cfd_amd/slab.py IpcCopyP2P: every rank copies its messages straight into its neighbours' receive buffers (IPC
    memory handles) and the ranks keep in step through the stream-ordered flags of include/lbm_abi.h
    (lbm_flag_post / lbm_flag_await) -- no RCCL kernel, no host staging.  Ranks share cuda:0 here; between GPUs the same
    calls are peer copies.  Must equal the undecomposed oracle bit for bit, and no await may have run into its time limit."""
    import torch.multiprocessing as mp
    world = len(splits)
    W, H, Dg = 136, 44, sum(splits)
    steps = (5, 4, 1, 3, 8)
    params = dict(inlet_velocity=0.06, omega_shear=1.7, omega_bulk=1.5)
    o = oracle.Solver(W, H, Dg, preset)
    o.init()
    o.add_solid(20, 0, 4, 9, 6, 25)
    o.add_solid(W - 2, 24, 9, 2, 5, 8)
    mask = o.solid_mask()
    o.set_params(**params)
    o.step(sum(steps))
    port = 29650 + (os.getpid() % 300) + 5 * world + preset
    out = mp.Manager().dict()
    mp.spawn(_rank, args=(world, port, preset, W, H, splits, steps, mask, params, out, "ipc"), nprocs=world, join=True)
    parts = [out[r] for r in range(world)]
    assert not any(p[6] for p in parts), "a flag wait ran into its time limit"
    rho = np.concatenate([p[2] for p in parts])
    u = np.concatenate([p[3] for p in parts])
    f = np.concatenate([p[4].reshape(19, p[1], H * W) for p in parts], axis=1).reshape(19, -1)
    fluid = mask == 0
    np.testing.assert_array_equal(rho, o.density())
    np.testing.assert_array_equal(u, o.velocity())
    np.testing.assert_array_equal(f[:, fluid], o.populations()[:, fluid])
