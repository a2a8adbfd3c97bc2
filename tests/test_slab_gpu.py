"""z-slab decomposition on ONE GPU: N slab contexts in one process exchange halo messages by
device copies (cfd_amd.slab.LocalSlabCluster, same packing / routing / schedule as the RCCL
path).  Every cell runs the same arithmetic whatever the decomposition, so the result must be
bit-identical to the undecomposed HIP run and to the oracle."""
import numpy as np
import pytest

from conftest import CUBE_OBJ

pytestmark = pytest.mark.gpu


def run_slabs(lbm, W, H, Dg, preset, splits, mask, steps, params=None, scheme=0):
    from cfd_amd import slab
    engines, z0 = [], 0
    for d in splits:
        engines.append(slab.HipSlabEngine(W, H, Dg, z0, d, preset, scheme=scheme))
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
    cl.step(steps)
    rho = np.concatenate([e.solver.density() for e in engines])
    u = np.concatenate([e.solver.velocity() for e in engines])
    f = np.concatenate([e.solver.populations().reshape(19, e.D, H * W) for e in engines], axis=1).reshape(19, -1)
    cls = np.concatenate([e.solver.classification() for e in engines])
    for e in engines:
        e.solver.close()
    return rho, u, f, cls


@pytest.mark.parametrize("scheme", [0, 1, 2])
@pytest.mark.parametrize("preset", [0, 1])
@pytest.mark.parametrize("splits", [(32,), (16, 16), (11, 10, 11), (1, 30, 1), (2, 2, 28)])
def test_slabs_equal_single_domain(oracle, lbm, preset, splits, scheme):
    W, H, Dg = 72, 24, 32
    o = oracle.Solver(W, H, Dg, preset)
    o.init()
    o.add_solid(20, 0, 4, 9, 11, 25)          # obstacle crossing every slab interface
    o.add_solid(50, 8, 14, 6, 6, 4)
    mask = o.solid_mask()
    o.step(9)
    rho, u, f, cls = run_slabs(lbm, W, H, Dg, preset, splits, mask, 9, scheme=scheme)
    np.testing.assert_array_equal(rho, o.density())
    np.testing.assert_array_equal(u, o.velocity())
    fluid = mask == 0
    np.testing.assert_array_equal(f[:, fluid], o.populations()[:, fluid])
    if scheme == 2:   # the sliding scheme carries the solid cells' frozen populations along
        np.testing.assert_array_equal(f, o.populations())
    np.testing.assert_array_equal(cls, o.classification()[0])


@pytest.mark.parametrize("scheme", [0, 1, 2])
def test_slabs_empty_tunnel_outlet_wrap(oracle, lbm, scheme):
    """No mesh: the outlet quirk rows at y=0 / y=H-1 and the z wrap between the first and the
    last layer cross slab interfaces and the ring link."""
    W, H, Dg = 40, 12, 12
    o = oracle.Solver(W, H, Dg)
    o.init(); o.step(15)
    mask = np.zeros(W * H * Dg, np.uint8)
    for splits in [(6, 6), (4, 4, 4), (12,)]:
        rho, u, f, _ = run_slabs(lbm, W, H, Dg, 0, splits, mask, 15, scheme=scheme)
        np.testing.assert_array_equal(rho, o.density(), err_msg=str(splits))
        np.testing.assert_array_equal(u, o.velocity(), err_msg=str(splits))
        np.testing.assert_array_equal(f, o.populations(), err_msg=str(splits))


@pytest.mark.parametrize("chunk", [0, 5])
@pytest.mark.parametrize("scheme", [0, 1, 2])
def test_slabs_cube_64(oracle, lbm, scheme, chunk, monkeypatch):
    """chunk 5: the interior update of every slab in launches of 5 layers (lbm_hip.hip auto_chunk)"""
    monkeypatch.setenv("LBM_AB_CHUNK", str(chunk))
    o = oracle.Solver(64, 64, 64)
    o.init(); o.load_mesh(CUBE_OBJ)
    mask = o.solid_mask()
    o.step(20)
    rho, u, _, _ = run_slabs(lbm, 64, 64, 64, 0, (16, 16, 16, 16), mask, 20, scheme=scheme)
    np.testing.assert_array_equal(rho, o.density())
    np.testing.assert_array_equal(u, o.velocity())
