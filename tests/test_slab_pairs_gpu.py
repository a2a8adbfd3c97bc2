"""Two steps per exchange pattern on z-slabs (include/lbm_abi.h lbm_pair_*): the fused kernel advances a slab's
interior by two steps while its two boundary layers and the rim take two ordinary steps, each followed by the usual
halo exchange.  Every cell must equal the undecomposed run / the oracle bit for bit, for any decomposition, both
presets, obstacles across the interfaces -- through the in-process cluster (device copies as transport) and
through SlabSolver's real two-stream schedule on a one-rank ring."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def build_cluster(lbm, W, H, Dg, preset, splits, mask, params=None):
    from cfd_amd import slab
    engines, z0 = [], 0
    for d in splits:
        engines.append(slab.HipSlabEngine(W, H, Dg, z0, d, preset))
        z0 += d
    assert z0 == Dg
    cl = slab.LocalSlabCluster(engines, preset)
    m3 = mask.reshape(Dg, H, W)
    for e in engines:
        e.solver.init()
        if params:
            e.solver.set_params(**params)
        e.solver.set_solid(m3[e.z0:e.z0 + e.D])
    cl.exchange_mask()
    cl.exchange_populations()
    return cl, engines


def gather(engines, H, W):
    rho = np.concatenate([e.solver.density() for e in engines])
    u = np.concatenate([e.solver.velocity() for e in engines])
    f = np.concatenate([e.solver.populations().reshape(19, e.D, H * W) for e in engines], axis=1).reshape(19, -1)
    return rho, u, f


@pytest.mark.parametrize("preset", [0, 1])
@pytest.mark.parametrize("splits", [(32,), (16, 16), (11, 10, 11), (8, 24), (7, 18, 7)])
def test_pairs_equal_oracle(oracle, lbm, preset, splits):
    if lbm.__name__ and __import__("os").environ.get("LBM_LAYOUT") != str(lbm.LAYOUT_ROWS):
        pytest.skip("the fused path runs on the row-interleaved layout")
    W, H, Dg = 136, 44, 32
    o = oracle.Solver(W, H, Dg, preset)
    o.init()
    o.add_solid(20, 0, 4, 9, 6, 25)           # obstacle crossing every slab interface
    o.add_solid(100, 16, 14, 6, 6, 4)
    o.add_solid(W - 2, 24, 9, 2, 5, 8)        # in the outlet column, across an interface
    mask = o.solid_mask()
    params = dict(inlet_velocity=0.06, omega_shear=1.7, omega_bulk=1.5)
    o.set_params(**params)
    cl, engines = build_cluster(lbm, W, H, Dg, preset, splits, mask, params)
    paired = cl.step_pairs(3)                   # thin or obstructed slabs decline and take single steps beside the others
    assert splits not in ((32,), (16, 16)) or all(paired)
    o.step(6)
    fluid = mask == 0
    f = np.concatenate([e.solver.populations().reshape(19, e.D, H * W) for e in engines], axis=1).reshape(19, -1)
    np.testing.assert_array_equal(f[:, fluid], o.populations()[:, fluid])
    if any(paired):
        with pytest.raises(lbm.LbmError) as err:   # the previous buffer is two steps old: no rho/u on demand
            engines[paired.index(True)].solver.density()
        assert err.value.code == -5
    cl.step(1); o.step(1)                       # a single split step brings them back
    cl.step_pairs(1); cl.step(2)
    o.step(4)
    rho, u, f = gather(engines, H, W)
    np.testing.assert_array_equal(rho, o.density())
    np.testing.assert_array_equal(u, o.velocity())
    np.testing.assert_array_equal(f[:, fluid], o.populations()[:, fluid])
    for e, p in zip(engines, paired):
        assert (e.solver.stats()["fused_cells"] > 0) == p
        e.solver.close()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("busy_streams", [0, 12])
def test_two_stream_schedule_with_pairs(lbm, busy_streams, monkeypatch):
    """SlabSolver.step(n) on a one-rank ring: (n - 1) // 2 pairs + single steps, boundary work and exchanges on the
    side stream beside the fused kernel; equals the plain solver"""
    import os
    import torch
    from cfd_amd import slab
    if os.environ.get("LBM_LAYOUT") != str(lbm.LAYOUT_ROWS):
        pytest.skip("the fused path runs on the row-interleaved layout")
    W, H, D = 136, 44, 40
    rects = [(20, 0, 3, 9, 6, 30), (134, 3, 10, 2, 4, 9)]
    g = lbm.Solver(W, H, D)
    g.init()
    for r in rects:
        g.add_solid(*r)
    others = [torch.cuda.Stream() for _ in range(busy_streams)]
    junk = []
    for st in others:
        with torch.cuda.stream(st):
            junk.append(torch.zeros(1 << 20, device="cuda").add_(1.0))
    e = slab.HipSlabEngine(W, H, D, 0, D, lbm.PRESET_WIND_TUNNEL, two_streams=True)
    drv = slab.SlabSolver(e, lbm.PRESET_WIND_TUNNEL, 1, 0, slab.LoopbackP2P())
    s = e.solver
    s.init()
    for r in rects:
        s.add_solid(*r)
    drv.exchange_mask()
    drv.exchange_populations()
    for n in (9, 4, 1, 6):
        g.step(n)
        drv.step(n)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(s.density(), g.density())
        np.testing.assert_array_equal(s.velocity(), g.velocity())
        np.testing.assert_array_equal(s.populations(), g.populations())
    assert s.stats()["fused_cells"] > 0
    # the same driver with pairs switched off takes single split steps only
    drv.use_pairs = False
    g.step(5); drv.step(5)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(s.populations(), g.populations())
    s.close(); g.close()


def test_pair_call_order_is_enforced(lbm):
    import os
    from cfd_amd import slab
    if os.environ.get("LBM_LAYOUT") != str(lbm.LAYOUT_ROWS):
        pytest.skip("the fused path runs on the row-interleaved layout")
    e = slab.HipSlabEngine(136, 44, 24, 0, 12, lbm.PRESET_WIND_TUNNEL)
    e.solver.init()
    send, recv = [e.new_message(), e.new_message()], [e.new_message(), e.new_message()]
    with pytest.raises(lbm.LbmError):
        e.pair_interior()
    e.pair_boundary_first(send[0], send[1])
    with pytest.raises(lbm.LbmError):
        e.solver.step(1)
    with pytest.raises(lbm.LbmError):
        e.halo_unpack_both(recv[0], recv[1])     # before lbm_pair_interior
    with pytest.raises(lbm.LbmError):
        e.pair_boundary_second(send[0], send[1])
    e.pair_interior()
    # round-2 advice: while a pair is in flight nothing but its own calls may touch the state -- the single-side
    # pack / unpack would write the ghost layers the fused kernel is reading, the mutators the mask and populations
    for call in (lambda: e.halo_unpack(0, recv[0]), lambda: e.halo_pack(1, send[1]), lambda: e.solver.add_solid(1, 1, 1, 2, 2, 2),
                 lambda: e.solver.init(), lambda: e.solver.set_solid(np.zeros(136 * 44 * 12, np.uint8)),
                 lambda: e.solver.set_populations(np.zeros((19, 136 * 44 * 12), np.float32)),
                 lambda: e.set_mask_ghost(0, np.zeros(136 * 44, np.uint8))):
        with pytest.raises(lbm.LbmError) as err:
            call()
        assert err.value.code == -5
    e.halo_unpack_both(recv[0], recv[1])
    e.pair_boundary_second(send[0], send[1])
    e.halo_unpack_both(recv[0], recv[1])
    e.pair_finish()
    e.solver.step(1)
    # ... and a split single step likewise keeps the mutators out until lbm_step_finish
    e.step_boundary(send[0], send[1])
    with pytest.raises(lbm.LbmError) as err:
        e.solver.add_solid(1, 1, 1, 2, 2, 2)
    assert err.value.code == -5
    e.step_interior()
    e.halo_pack(0, send[0])                      # (packing the new state before lbm_step_finish is part of the protocol)
    e.step_finish()
    e.solver.add_solid(1, 1, 1, 2, 2, 2)
    e.solver.close()
    # an undecomposed context has no pairs of this kind
    g = lbm.Solver(136, 30, 24)
    g.init()
    import ctypes as C
    a = C.c_int(1)
    assert g._lib.lbm_pair_available(g._ctx, C.byref(a)) == 0 and a.value == 0
    assert g._lib.lbm_pair_boundary_first(g._ctx, None, None) == -5
