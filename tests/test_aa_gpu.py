"""AA-pattern (in-place, one buffer) variant against the CPU oracle and against the A/B scheme:
identical arithmetic per cell, so rho / u / fluid populations must be bit-identical after any
number of steps, even or odd."""
import numpy as np
import pytest

from conftest import CUBE_OBJ

pytestmark = pytest.mark.gpu


def pair(oracle, lbm, W, H, D, preset=0):
    return oracle.Solver(W, H, D, preset), lbm.Solver(W, H, D, preset, scheme=lbm.SCHEME_AA)


def same(o, g, what=""):
    np.testing.assert_array_equal(g.solid_mask() > 0, o.solid_mask() > 0, err_msg=what + " mask")
    np.testing.assert_array_equal(g.density(), o.density(), err_msg=what + " rho")
    np.testing.assert_array_equal(g.velocity(), o.velocity(), err_msg=what + " u")
    fluid = o.solid_mask() == 0
    np.testing.assert_array_equal(g.populations()[:, fluid], o.populations()[:, fluid], err_msg=what + " f")


@pytest.mark.parametrize("nsteps", [1, 2, 9, 10, 101])
def test_aa_config1_cube_64(oracle, lbm, nsteps):
    o, g = pair(oracle, lbm, 64, 64, 64)
    o.init(); g.init()
    o.load_mesh(CUBE_OBJ); g.load_mesh(CUBE_OBJ)
    o.step(nsteps); g.step(nsteps)
    same(o, g, f"N={nsteps}")
    np.testing.assert_array_equal(g.curl_field(), (o.compute_curl(), o.curl_field())[1])


@pytest.mark.parametrize("dims", [(32, 32, 32), (70, 33, 17), (1, 5, 7), (130, 4, 3), (64, 2, 1), (5, 1, 1), (62, 9, 8), (63, 9, 8), (94, 3, 5)])
@pytest.mark.parametrize("preset", [0, 1])
def test_aa_ragged_domains(oracle, lbm, dims, preset):
    """incl. W = 62, 63, 94: ghost slots x = -1 / x = W share the row padding"""
    W, H, D = dims
    o, g = pair(oracle, lbm, W, H, D, preset)
    o.init(); g.init()
    o.step(3); g.step(3)                               # odd count: buffer left in the swapped layout
    same(o, g, "empty")
    for s in (o, g):
        s.add_solid(W // 3, 0, D // 4, max(1, W // 5), max(1, H // 2), max(1, D // 2))
        s.add_solid(W - 2, H - 1, 0, 5, 5, 1)
    o.step(4); g.step(4)
    same(o, g, "rects after odd phase")
    o.step(3); g.step(3)
    same(o, g, "rects")


def test_aa_step_by_step_and_params(oracle, lbm):
    """single-step calls (moments every step), live parameter changes, reset, write_populations"""
    rng = np.random.default_rng(3)
    W, H, D = 96, 24, 20
    o, g = pair(oracle, lbm, W, H, D)
    mask = (rng.random(W * H * D) < 0.08).astype(np.uint8) * 255
    o.init(); g.init()
    o.set_solid(mask); g.set_solid(mask)
    for i in range(5):
        o.step(1); g.step(1)
        same(o, g, f"step {i}")
    for s in (o, g):
        s.set_params(inlet_velocity=0.05, omega_shear=1.7, omega_bulk=1.2)
    o.step(5); g.step(5)
    same(o, g, "changed params")
    o.reset(); g.reset()
    o.step(2); g.step(2)
    same(o, g, "after reset")
    f = o.populations()
    g.set_populations(f)
    np.testing.assert_array_equal(g.populations(), f)
    o.step(3); g.step(3)
    same(o, g, "after import")


def test_aa_remesh_after_steps(oracle, lbm):
    """mask replaced mid-run: cells solid since init() become fluid again (same as the reference
    because their frozen populations are still the equilibrium init() wrote)"""
    o, g = pair(oracle, lbm, 64, 32, 32)
    o.init(); g.init()
    o.add_solid(10, 5, 5, 8, 8, 8); g.add_solid(10, 5, 5, 8, 8, 8)
    o.add_solid(60, 3, 3, 4, 9, 9); g.add_solid(60, 3, 3, 4, 9, 9)     # touches the outlet column
    o.step(5); g.step(5)
    o.load_mesh(CUBE_OBJ); g.load_mesh(CUBE_OBJ)
    o.step(6); g.step(6)
    same(o, g, "after remesh")


def test_aa_equals_ab_and_halves_memory(lbm):
    a = lbm.Solver(128, 64, 64, preset=lbm.PRESET_CAVITY)
    b = lbm.Solver(128, 64, 64, preset=lbm.PRESET_CAVITY, scheme=lbm.SCHEME_AA)
    for s in (a, b):
        s.init(); s.add_solid(40, 20, 20, 10, 10, 10); s.step(25)
    np.testing.assert_array_equal(a.density(), b.density())
    np.testing.assert_array_equal(a.velocity(), b.velocity())
    assert b.stats()["population_bytes"] < 0.7 * a.stats()["population_bytes"]   # one buffer + ghost shell padding
