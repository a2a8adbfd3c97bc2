"""The z-slab halo protocol (cfd_amd/slab.py: neighbour topology, message routing, schedule) on
CPU: oracle-backed slab engines, (a) inside one process via LocalSlabCluster and (b) as two real
ranks over torch.distributed/gloo via SlabSolver + TorchP2P -- the same driver code bench.py runs
over RCCL.  A decomposed run must equal the undecomposed oracle bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests"))


def reference_run(oracle, W, H, Dg, preset, steps, rects):
    o = oracle.Solver(W, H, Dg, preset)
    o.init()
    for r in rects:
        o.add_solid(*r)
    mask = o.solid_mask()
    o.step(steps)
    return o, mask


RECTS = [(8, 0, 2, 5, 6, 9), (20, 5, 5, 3, 3, 2)]


@pytest.mark.parametrize("preset", [0, 1])
@pytest.mark.parametrize("splits", [(12,), (6, 6), (5, 3, 4), (1, 10, 1)])
def test_local_cluster_equals_single_domain(oracle, preset, splits):
    from cfd_amd import slab
    from oracle_slab import OracleSlabEngine
    W, H, Dg, steps = 28, 10, 12, 7
    o, mask = reference_run(oracle, W, H, Dg, preset, steps, RECTS)
    engines, z0 = [], 0
    for d in splits:
        e = OracleSlabEngine(W, H, Dg, z0, d, preset)
        e.set_solid(mask.reshape(Dg, H, W)[z0:z0 + d])
        engines.append(e)
        z0 += d
    cl = slab.LocalSlabCluster(engines, preset)
    cl.exchange_mask()
    cl.exchange_populations()
    cl.step(steps)
    rho = np.concatenate([e.rho for e in engines])
    u = np.concatenate([e.u for e in engines]).reshape(-1, 4)
    np.testing.assert_array_equal(rho, o.density())
    np.testing.assert_array_equal(u, o.velocity())
    f = np.concatenate([e.populations().reshape(19, e.D, -1) for e in engines], axis=1).reshape(19, -1)
    np.testing.assert_array_equal(f, o.populations())


def test_empty_tunnel_wrap_rows(oracle):
    """No obstacle: the outlet quirk rows (y = 0, y = H-1) and the first<->last layer wrap are live."""
    from cfd_amd import slab
    from oracle_slab import OracleSlabEngine
    W, H, Dg, steps = 16, 6, 6, 9
    o, mask = reference_run(oracle, W, H, Dg, 0, steps, [])
    for splits in [(3, 3), (2, 2, 2), (6,)]:
        engines, z0 = [], 0
        for d in splits:
            engines.append(OracleSlabEngine(W, H, Dg, z0, d, 0))
            z0 += d
        cl = slab.LocalSlabCluster(engines, 0)
        cl.exchange_mask(); cl.exchange_populations(); cl.step(steps)
        f = np.concatenate([e.populations().reshape(19, e.D, -1) for e in engines], axis=1).reshape(19, -1)
        np.testing.assert_array_equal(f, o.populations(), err_msg=str(splits))


def _gloo_rank(rank, world, port, preset, W, H, Dg, steps, mask, out):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cfd_amd import slab
    from oracle_slab import OracleSlabEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z0, d = slab.slab_range(Dg, world, rank)
    e = OracleSlabEngine(W, H, Dg, z0, d, preset)
    e.set_solid(mask.reshape(Dg, H, W)[z0:z0 + d])
    drv = slab.SlabSolver(e, preset, world, rank, slab.TorchP2P())
    drv.exchange_mask()
    drv.exchange_populations()
    drv.step(steps)
    out[rank] = (z0, d, e.rho.copy(), e.u.copy(), e.populations())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("preset,world", [(0, 2), (1, 2), (0, 3)])
def test_gloo_ranks_equal_single_domain(oracle, preset, world):
    import torch.multiprocessing as mp
    W, H, Dg, steps = 24, 8, 9, 6
    o, mask = reference_run(oracle, W, H, Dg, preset, steps, [(6, 0, 1, 5, 4, 7)])
    port = 29600 + (os.getpid() % 200) + preset * 7 + world
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gloo_rank, args=(world, port, preset, W, H, Dg, steps, mask, out), nprocs=world, join=True)
    parts = [out[r] for r in range(world)]
    rho = np.concatenate([p[2] for p in parts])
    u = np.concatenate([p[3] for p in parts]).reshape(-1, 4)
    f = np.concatenate([p[4].reshape(19, p[1], -1) for p in parts], axis=1).reshape(19, -1)
    np.testing.assert_array_equal(rho, o.density())
    np.testing.assert_array_equal(u, o.velocity())
    np.testing.assert_array_equal(f, o.populations())


def _gloo_pair_rank(rank, world, port, preset, W, H, splits, steps, pairing, mask, out):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cfd_amd import slab
    from oracle_slab import OracleSlabEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Dg = sum(splits)
    z0, d = sum(splits[:rank]), splits[rank]
    e = OracleSlabEngine(W, H, Dg, z0, d, preset, two_streams=True, pairs=pairing[rank])
    e.set_solid(mask.reshape(Dg, H, W)[z0:z0 + d])
    drv = slab.SlabSolver(e, preset, world, rank, slab.TorchP2P())
    drv.exchange_mask()
    drv.exchange_populations()
    for n in steps:                                    # several step(n) calls, odd and even n
        drv.step(n)
    out[rank] = (z0, d, e.rho.copy(), e.u.copy(), e.populations(), list(e.log))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("preset,splits,pairing", [(0, (5, 4), (True, True)), (0, (5, 4), (True, False)), (1, (4, 5), (False, True)),
                                                   (0, (3, 3, 3), (True, False, True)), (1, (2, 4, 3), (False, True, True))])
def test_gloo_pairs_of_steps(oracle, preset, splits, pairing):
    """SlabSolver.step(n) across real processes with the PAIR schedule (lbm_pair_*: two steps per exchange pattern):
    ranks that pair next to ranks that decline, odd and even n, must equal the undecomposed oracle bit for bit --
    a pair sends what two single split steps send, in the same order (round-2 review: this ran in-process only)."""
    import torch.multiprocessing as mp
    world = len(splits)
    W, H, Dg = 24, 8, sum(splits)
    steps = (5, 4, 1, 3)
    o, mask = reference_run(oracle, W, H, Dg, preset, sum(steps), [(6, 0, 1, 5, 4, 7)])
    port = 29850 + (os.getpid() % 100) + 3 * world + preset + 11 * sum(pairing)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gloo_pair_rank, args=(world, port, preset, W, H, splits, steps, pairing, mask, out), nprocs=world, join=True)
    parts = [out[r] for r in range(world)]
    for r, p in enumerate(parts):
        # step(5): 2 pairs + 1 single; step(4): 1 pair + 2 singles; step(1): 1 single; step(3): 1 pair + 1 single
        want = ["pair", "pair", "single", "pair", "single", "single", "single", "pair", "single"] if pairing[r] else ["single"] * sum(steps)
        assert p[5] == want, (r, p[5])
    rho = np.concatenate([p[2] for p in parts])
    u = np.concatenate([p[3] for p in parts]).reshape(-1, 4)
    f = np.concatenate([p[4].reshape(19, p[1], -1) for p in parts], axis=1).reshape(19, -1)
    np.testing.assert_array_equal(rho, o.density())
    np.testing.assert_array_equal(u, o.velocity())
    np.testing.assert_array_equal(f, o.populations())


def test_balanced_ranges():
    """slabs re-cut for equal measured cost (bench.py strong scaling): a partition of the depth, thinner where a
    layer costs more, never under the minimum"""
    from cfd_amd import slab
    even = [slab.slab_range(512, 8, r) for r in range(8)]
    times = [0.691, 0.737, 0.613, 0.611, 0.608, 0.619, 0.722, 0.689]       # the eight slabs of configs[3], ms per step
    cut = slab.balanced_ranges(512, even, times)
    assert [z for z, _ in cut] == [0] + list(np.cumsum([d for _, d in cut])[:-1]) and sum(d for _, d in cut) == 512
    cost = np.concatenate([[t / d] * d for (_, d), t in zip(even, times)])
    new = [cost[z:z + d].sum() for z, d in cut]
    assert max(new) < 0.67 < max(times) and max(new) - min(new) < 0.02
    assert cut[1][1] < 64 < cut[3][1]
    assert slab.balanced_ranges(512, even, [1.0] * 8) == even
    assert slab.balanced_ranges(40, [(0, 20), (20, 20)], [100.0, 1.0], min_layers=8) == [(0, 10), (10, 30)]
    assert slab.balanced_ranges(20, [(0, 10), (10, 10)], [100.0, 1.0], min_layers=8) == [(0, 8), (8, 12)]
    with pytest.raises(ValueError):
        slab.balanced_ranges(512, even, times[:-1])


def test_topology():
    from cfd_amd import slab
    assert [slab.slab_range(10, 3, r) for r in range(3)] == [(0, 4), (4, 3), (7, 3)]
    assert slab.neighbours(0, 4, 0) == (3, 1) and slab.neighbours(0, 4, 3) == (2, 0)   # ring
    assert slab.neighbours(1, 4, 0) == (None, 1) and slab.neighbours(1, 4, 3) == (2, None)  # chain
    assert slab.neighbours(1, 1, 0) == (None, None) and slab.neighbours(0, 1, 0) == (0, 0)


def test_balance_slabs_two_passes():
    """bench.py balance_slabs: re-cut, measure, re-cut from that measurement; stops when a partition repeats"""
    import importlib.util
    from cfd_amd import slab
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    depth, P = 512, 8
    layer_cost = np.where((np.arange(depth) > 150) & (np.arange(depth) < 360), 0.8, 1.0)     # a lighter middle (the car)

    def times_of(ranges):      # cost of a slab, with a cliff: beyond 66 layers a slab pays for one more round of blocks
        return [float(layer_cost[z0:z0 + d].sum()) * (1.08 if d > 66 else 1.0) for z0, d in ranges]
    calls = []

    def measure(cut):
        calls.append(cut)
        return times_of(cut), {"cut": cut}
    even = [slab.slab_range(depth, P, r) for r in range(P)]
    tried = bench.balance_slabs(slab, depth, even, times_of(even), measure)
    assert 1 <= len(tried) <= 2 and len(calls) == len(tried)
    for cut, slowest, res in tried:
        assert sum(d for _, d in cut) == depth and cut[0][0] == 0 and res["cut"] == cut
        assert slowest == max(times_of(cut))
    assert min(t[1] for t in tried) < max(times_of(even))
    # a uniform domain: the first re-cut is the even split again, nothing to try
    assert bench.balance_slabs(slab, depth, even, [1.0] * P, measure) == []
