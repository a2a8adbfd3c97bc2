"""BASELINE.json configs[2] and configs[3] at FULL size, where the CPU oracle would take minutes:
size-independent properties instead.  Every cell runs the same fp32 arithmetic whatever the scheme
or the decomposition, so at any size (i) the in-place AA kernel must reproduce the A/B fields bit
for bit, and so must the sliding in-place scheme, (ii) a z-slab run must reproduce the undecomposed run bit for bit, and (iii) the device
voxeliser must reproduce the host voxeliser's mask.  The small twins of these cases are checked
against the oracle itself (tests/test_parity_gpu.py, tests/golden/oracle_cases.json)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
CAR_OBJ = os.path.join(GOLDEN, "car_like.obj")


@pytest.fixture
def lbm_rows(monkeypatch):
    from cfd_amd import lbm as L
    L.load_library()
    monkeypatch.delenv("LBM_LAYOUT", raising=False)
    return L


def _run(L, dims, scheme, steps):
    s = L.Solver(*dims, scheme=scheme)
    s.init()
    s.load_mesh(CAR_OBJ)
    s.step(steps)
    return s


def test_config2_512x256x256_car(lbm_rows):
    L = lbm_rows
    dims, steps = (512, 256, 256), 21
    a = _run(L, dims, L.SCHEME_AB, steps)
    mask = a.solid_mask()
    host, _, rc = L.host_voxelize_obj(CAR_OBJ, *dims)
    assert rc == 0
    np.testing.assert_array_equal(mask, host)
    rho, u = a.density(), a.velocity()
    a.close()
    fluid = mask == 0
    assert np.isfinite(rho).all() and np.isfinite(u).all()
    assert np.all(rho[~fluid] == 0) and rho[fluid].min() > 0.5 and rho[fluid].max() < 1.5
    b = _run(L, dims, L.SCHEME_AA, steps)
    np.testing.assert_array_equal(b.density(), rho)
    np.testing.assert_array_equal(b.velocity(), u)
    b.close()
    # two z-slabs on this GPU, device copies as the transport
    from cfd_amd import slab
    W, H, Dg = dims
    engines = [slab.HipSlabEngine(W, H, Dg, z0, Dg // 2, L.PRESET_WIND_TUNNEL) for z0 in (0, Dg // 2)]
    cl = slab.LocalSlabCluster(engines, L.PRESET_WIND_TUNNEL)
    for e in engines:
        e.solver.init()
        e.solver.load_mesh(CAR_OBJ)
    cl.exchange_mask(); cl.exchange_populations(); cl.step(steps)
    np.testing.assert_array_equal(np.concatenate([e.solver.density() for e in engines]), rho)
    np.testing.assert_array_equal(np.concatenate([e.solver.velocity() for e in engines]), u)
    for e in engines:
        e.solver.close()


def test_config3_1024x512x512_car_schemes_agree(lbm_rows):
    """268 M cells: beyond the reference's 2^31 index limit for 19 n."""
    L = lbm_rows
    dims, steps = (1024, 512, 512), 5
    a = _run(L, dims, L.SCHEME_AB, steps)
    a.set_field_window(200, 112)                       # the layers around the car are enough to compare
    rho, u = a.density(), a.velocity()
    nsolid = int((a.solid_mask() > 0).sum())
    a.close()
    assert np.isfinite(rho).all() and nsolid > 1024 * 512
    b = L.Solver(*dims, scheme=L.SCHEME_AA)
    b.set_field_window(200, 112)
    b.init(); b.load_mesh(CAR_OBJ); b.step(steps)
    np.testing.assert_array_equal(b.density(), rho)
    np.testing.assert_array_equal(b.velocity(), u)
    assert int((b.solid_mask() > 0).sum()) == nsolid
    b.close()
    c = L.Solver(*dims, scheme=L.SCHEME_SLIDE)          # 32 launches per step, solid cells carried along
    c.set_field_window(200, 112)
    c.init(); c.load_mesh(CAR_OBJ); c.step(steps)
    np.testing.assert_array_equal(c.density(), rho)
    np.testing.assert_array_equal(c.velocity(), u)
    c.close()


def test_config4_hbm_sized_in_place(lbm_rows):
    """BASELINE.json configs[4]: one buffer sized to HBM (3.43 G cells, 267 GB; more floats than one
    grid dimension can index, more cells than the reference's int index can), with both in-place
    schemes.  Properties: in the empty wind tunnel only the inlet disturbs the rest state (walls,
    free-slip faces and the outlet copy map rest onto rest), so every cell further from the x ends
    than the steps taken must still hold exactly the rest value a small domain shows -- at the
    bottom, the middle and the top of the 1484 layers, which an unfilled or mis-addressed part of the
    buffer would not -- while the inlet column must not; and the two schemes, which share no
    addressing code (AA: one buffer in two layouts; sliding: the A/B kernel between two positions of
    one buffer, 371 launches per step), must agree on every value of every window bit for bit."""
    import torch
    L = lbm_rows
    free, _ = torch.cuda.mem_get_info()
    dims, steps = (1504, 1536, 1484), 4
    need = 19 * 1536 * 4 * 1538 * 1486 + (1 << 30)
    if free < need:
        pytest.skip(f"needs {need >> 30} GiB of free HBM, {free >> 30} GiB available")
    small = L.Solver(64, 64, 64, scheme=L.SCHEME_AA)
    small.init(); small.step(steps)
    rest = small.density().reshape(64, 64, 64)[32, 32, 32]
    small.close()
    W, H, D = dims
    seen = {}
    for scheme in (L.SCHEME_AA, L.SCHEME_SLIDE):
        g = L.Solver(*dims, scheme=scheme)
        assert g.stats()["population_bytes"] < 270e9
        g.init()
        for z0 in (0, 740, D - 6):
            g.set_field_window(z0, 6)
            g.step(steps)
            rho = g.density().reshape(6, H, W)
            u = g.velocity().reshape(6, H, W, 4)
            assert np.isfinite(rho).all() and np.isfinite(u).all()
            zs = slice(0, 6) if 0 < z0 < D - 6 else (slice(5, 6) if z0 == 0 else slice(0, 1))
            inner = rho[zs, 40:H - 40, 40:W - 40]
            np.testing.assert_array_equal(inner, np.full_like(inner, rest))
            assert np.all(u[zs, 40:H - 40, 40:W - 40, :3] == 0)
            assert rho[zs, 40:H - 40, 0].max() > rest      # the inlet pushes
            if scheme == L.SCHEME_AA:
                seen[z0] = (rho.copy(), u.copy())
            else:
                np.testing.assert_array_equal(rho, seen[z0][0], err_msg=f"sliding vs AA, rho, window at {z0}")
                np.testing.assert_array_equal(u, seen[z0][1], err_msg=f"sliding vs AA, u, window at {z0}")
        g.close()
