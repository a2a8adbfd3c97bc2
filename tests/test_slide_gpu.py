"""Sliding in-place scheme (LBM_SCHEME_SLIDE: the A/B kernel on one buffer plus chunk + 2 spare
layers, include/lbm_abi.h) against the CPU oracle and against the A/B scheme.  Same arithmetic per
cell and a solid cell's frozen populations move with the lattice, so EVERY result -- rho, u, curl and
all 19 populations of every cell, solid ones included -- must be bit-identical after any number of
steps, for any chunk size (1 layer per launch up to the whole domain)."""
import numpy as np
import pytest

from conftest import CUBE_OBJ

pytestmark = pytest.mark.gpu


def pair(oracle, lbm, W, H, D, preset=0, chunk=0):
    return oracle.Solver(W, H, D, preset), lbm.Solver(W, H, D, preset, scheme=lbm.SCHEME_SLIDE, slide_chunk=chunk)


def same(o, g, what=""):
    np.testing.assert_array_equal(g.solid_mask() > 0, o.solid_mask() > 0, err_msg=what + " mask")
    np.testing.assert_array_equal(g.density(), o.density(), err_msg=what + " rho")
    np.testing.assert_array_equal(g.velocity(), o.velocity(), err_msg=what + " u")
    np.testing.assert_array_equal(g.populations(), o.populations(), err_msg=what + " f")


@pytest.mark.parametrize("nsteps", [1, 2, 9, 10, 101])
def test_slide_config1_cube_64(oracle, lbm, nsteps):
    o, g = pair(oracle, lbm, 64, 64, 64)
    o.init(); g.init()
    o.load_mesh(CUBE_OBJ); g.load_mesh(CUBE_OBJ)
    o.step(nsteps); g.step(nsteps)
    same(o, g, f"N={nsteps}")
    np.testing.assert_array_equal(g.curl_field(), (o.compute_curl(), o.curl_field())[1])


@pytest.mark.parametrize("chunk", [1, 2, 3, 5, 7, 8, 64])
@pytest.mark.parametrize("preset", [0, 1])
def test_slide_chunk_sizes(oracle, lbm, chunk, preset):
    """chunks that divide the depth, that do not, one layer per launch, more layers than the domain"""
    W, H, D = 70, 12, 16
    o, g = pair(oracle, lbm, W, H, D, preset, chunk)
    o.init(); g.init()
    for s in (o, g):
        s.add_solid(20, 0, 3, 9, 7, 9)
        s.add_solid(W - 3, 2, D - 2, 3, 5, 2)          # outlet column, top layers (quirk Q2 wraps to z = 0)
    o.step(5); g.step(5)                               # odd count: lattice left at the shifted offset
    same(o, g, f"chunk {chunk}, 5 steps")
    o.step(4); g.step(4)
    same(o, g, f"chunk {chunk}, 9 steps")


@pytest.mark.parametrize("dims", [(32, 32, 32), (70, 33, 17), (1, 5, 7), (130, 4, 3), (64, 2, 1), (5, 1, 1), (63, 9, 8), (94, 3, 40)])
@pytest.mark.parametrize("preset", [0, 1])
def test_slide_ragged_domains(oracle, lbm, dims, preset):
    W, H, D = dims
    o, g = pair(oracle, lbm, W, H, D, preset)
    o.init(); g.init()
    o.step(3); g.step(3)
    same(o, g, "empty")
    for s in (o, g):
        s.add_solid(W // 3, 0, D // 4, max(1, W // 5), max(1, H // 2), max(1, D // 2))
        s.add_solid(W - 2, H - 1, 0, 5, 5, 1)
    o.step(4); g.step(4)
    same(o, g, "rects after odd phase")
    o.step(3); g.step(3)
    same(o, g, "rects")


def test_slide_step_by_step_and_params(oracle, lbm):
    """single-step calls (moments every step), live parameter changes, reset, write_populations"""
    rng = np.random.default_rng(3)
    W, H, D = 96, 24, 20
    o, g = pair(oracle, lbm, W, H, D, chunk=6)
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


def test_slide_remesh_mid_run(oracle, lbm):
    """mask replaced mid-run, at an odd and at an even step count: cells that were FLUID for a while,
    then solid, then fluid again keep their frozen populations as in the reference (unlike AA)"""
    o, g = pair(oracle, lbm, 64, 32, 32, chunk=5)
    o.init(); g.init()
    o.step(3); g.step(3)
    o.add_solid(10, 5, 5, 8, 8, 8); g.add_solid(10, 5, 5, 8, 8, 8)
    o.add_solid(60, 3, 3, 4, 9, 9); g.add_solid(60, 3, 3, 4, 9, 9)     # touches the outlet column
    o.step(5); g.step(5)
    same(o, g, "rects")
    o.load_mesh(CUBE_OBJ); g.load_mesh(CUBE_OBJ)                       # the rects turn fluid again
    o.step(6); g.step(6)
    same(o, g, "after remesh")
    rng = np.random.default_rng(11)
    mask = (rng.random(64 * 32 * 32) < 0.1).astype(np.uint8) * 255
    o.set_solid(mask); g.set_solid(mask)
    o.step(3); g.step(3)
    same(o, g, "random mask")


def test_slide_field_window(oracle, lbm):
    W, H, D = 64, 16, 24
    o, g = pair(oracle, lbm, W, H, D, chunk=7)
    o.init(); g.init()
    o.add_solid(20, 0, 8, 6, 9, 9); g.add_solid(20, 0, 8, 6, 9, 9)
    g.set_field_window(5, 9)
    o.step(7); g.step(7)
    np.testing.assert_array_equal(g.density(), o.density().reshape(D, H, W)[5:14].ravel())
    np.testing.assert_array_equal(g.velocity(), o.velocity().reshape(D, H, W, -1)[5:14].reshape(g.velocity().shape))
    g.set_field_window(0, 0)
    o.step(2); g.step(2)
    with pytest.raises(lbm.LbmError):
        g.density()
    np.testing.assert_array_equal(g.populations(), o.populations())


def test_slide_equals_ab_and_memory(lbm):
    a = lbm.Solver(128, 64, 64, preset=lbm.PRESET_CAVITY)
    b = lbm.Solver(128, 64, 64, preset=lbm.PRESET_CAVITY, scheme=lbm.SCHEME_SLIDE, slide_chunk=8)
    for s in (a, b):
        s.init(); s.add_solid(40, 20, 20, 10, 10, 10); s.step(25)
    np.testing.assert_array_equal(a.density(), b.density())
    np.testing.assert_array_equal(a.velocity(), b.velocity())
    np.testing.assert_array_equal(a.populations(), b.populations())
    assert b.stats()["population_bytes"] < 0.6 * a.stats()["population_bytes"]   # (64 + 2 + 10) / (2 * 66) layers


@pytest.mark.parametrize("chunk", [1, 2, 5])
@pytest.mark.parametrize("splits", [(7, 9), (1, 14, 1), (2, 3, 11), (16,)])
def test_slide_slabs_chunks(oracle, lbm, splits, chunk):
    """z-slabs of the sliding scheme (the boundary layer the sweep reaches last sends its message from
    a message-only pass): odd and even step counts, slabs of one and two layers, small chunks"""
    from test_slab_gpu import run_slabs
    W, H, Dg = 72, 10, 16
    for preset in (0, 1):
        o = oracle.Solver(W, H, Dg, preset)
        o.init()
        o.add_solid(20, 0, 2, 9, 6, 13)
        o.add_solid(W - 2, 3, 6, 2, 4, 5)          # solid cells in the outlet column across interfaces
        mask = o.solid_mask()
        o.step(7)
        import os
        os.environ["LBM_SLIDE_CHUNK"] = str(chunk)
        try:
            rho, u, f, _ = run_slabs(lbm, W, H, Dg, preset, splits, mask, 7, scheme=lbm.SCHEME_SLIDE)
        finally:
            del os.environ["LBM_SLIDE_CHUNK"]
        np.testing.assert_array_equal(rho, o.density(), err_msg=f"preset {preset}")
        np.testing.assert_array_equal(u, o.velocity(), err_msg=f"preset {preset}")
        np.testing.assert_array_equal(f, o.populations(), err_msg=f"preset {preset}")


def test_slide_slab_unpack_order_enforced(lbm):
    """the ghost layers of the new lattice position are input until the interior has run: the
    library refuses an unpack between lbm_step_boundary_layers and lbm_step_interior_layers"""
    from cfd_amd import slab
    e = slab.HipSlabEngine(64, 8, 12, 0, 6, lbm.PRESET_WIND_TUNNEL, scheme=lbm.SCHEME_SLIDE)
    e.solver.init()
    send, recv = [e.new_message(), e.new_message()], [e.new_message(), e.new_message()]
    for m in recv:
        m.zero_()
    e.step_boundary(send[0], send[1])
    with pytest.raises(lbm.LbmError):
        e.halo_unpack_both(recv[0], recv[1])
    with pytest.raises(lbm.LbmError):
        e.solver.step(1)                      # a split step is in flight
    e.step_interior()
    e.halo_unpack_both(recv[0], recv[1])
    e.step_finish()
    e.solver.step(1)
    e.solver.close()


def test_slide_pristine_solids(oracle, lbm):
    """the engine's call order (init, mask, steps): all solid cells hold w_k, so the sliding scheme
    skips them instead of carrying them along; read-backs, a later mask change (solid cells turning
    fluid again, fluid cells turning solid), reset and population import must all still equal the
    oracle, solid cells included"""
    W, H, D = 200, 20, 24
    o, g = pair(oracle, lbm, W, H, D, chunk=5)
    o.init(); g.init()
    for s in (o, g):
        s.add_solid(30, 0, 4, 90, 12, 14)              # all-solid 64-cell tiles inside
        s.add_solid(W - 70, 5, 10, 70, 6, 6)           # reaches the outlet column, fills the last tile of its rows
    o.step(5); g.step(5)
    same(o, g, "pristine, 5 steps")
    o.step(4); g.step(4)
    same(o, g, "pristine, 9 steps (after a read-back)")
    o.load_mesh(CUBE_OBJ); g.load_mesh(CUBE_OBJ)       # the boxes turn fluid, the cube and the floor solid
    o.step(6); g.step(6)
    same(o, g, "after the mask change")
    o.step(3); g.step(3)
    same(o, g, "carried along from here on")
    o.reset(); g.reset()                               # pristine again, mask kept
    o.step(7); g.step(7)
    same(o, g, "after reset")
    f = o.populations()
    g.set_populations(f)
    o.step(2); g.step(2)
    same(o, g, "after import")
