"""Two steps per pass (cfd_amd/csrc/k_fused2.h, lbm_opts.two_step): lbm_step(ctx, n >= 3, or n = 2 when the caller did
not read rho/u after its previous call) on an A/B context advances blocks of plain-fluid tiles by TWO time steps in one kernel and takes the rim (walls, tiles near
solids, ragged remainders) through two ordinary steps via a third buffer.  Every cell must come out
bit-identical to the CPU oracle's single steps (lbm_solver.cu:242-402, 477-494) -- fields, populations and
frozen solid cells -- for both presets, every compiled kernel variant, any z split, ragged sizes, obstacles,
and across mask / parameter changes between calls."""
import numpy as np
import pytest

from conftest import CUBE_OBJ

pytestmark = pytest.mark.gpu

VARIANTS = ["12,1,1", "9,1,2", "13,1,1", "9,0,2", "12,0,2"]     # the first one is the library default


def same(o, g, what=""):
    np.testing.assert_array_equal(g.density(), o.density(), err_msg=what + " rho")
    np.testing.assert_array_equal(g.velocity(), o.velocity(), err_msg=what + " u")
    np.testing.assert_array_equal(g.populations(), o.populations(), err_msg=what + " f")


def fused_in_use(lbm, g, may_decline=False):
    """the fused path runs on the row-interleaved layout only; it may decline a domain of which it would
    cover less than a quarter (one z chunk per column around obstacles)"""
    import os
    rows = os.environ.get("LBM_LAYOUT") == str(lbm.LAYOUT_ROWS)
    used = g.stats()["fused_cells"] > 0
    assert not used or rows
    assert used or not rows or may_decline
    return used


@pytest.mark.parametrize("preset", [0, 1])
@pytest.mark.parametrize("nsteps", [2, 3, 4, 11, 60])
def test_fused_empty_box(oracle, lbm, preset, nsteps):
    """no obstacle: walls, inlet / outlet (or cavity walls + lid) only; W ragged, last y tile short"""
    W, H, D = 136, 27, 22
    o = oracle.Solver(W, H, D, preset)
    g = lbm.Solver(W, H, D, preset)
    o.init(); g.init()
    o.step(nsteps); g.step(nsteps)
    same(o, g, f"preset {preset}, {nsteps} steps")
    fused_in_use(lbm, g)
    # the caller has read rho/u now: a call of one or two steps writes them as it goes and cannot fuse
    g2 = lbm.Solver(W, H, D, preset)
    g2.init()
    g2.density()
    g2.step(2)
    assert g2.stats()["fused_cells"] == 0         # nothing prepared for calls that cannot fuse
    g2.close()


@pytest.mark.parametrize("preset", [0, 1])
def test_even_batches_end_on_a_pair(oracle, lbm, preset):
    """an even lbm_step(n) whose caller did not read rho/u before runs in pairs to the end (no single step behind the
    last pair); rho/u asked for afterwards are those of the LAST step all the same (one step from the buffer two steps
    back into the rim's buffer, then the gather), also when a mask change settles them first, when the caller changed
    the parameters in between, and over several such calls in a row"""
    import os
    rows = os.environ.get("LBM_LAYOUT") == str(lbm.LAYOUT_ROWS)
    W, H, D = 136, 30, 26
    o = oracle.Solver(W, H, D, preset)
    g = lbm.Solver(W, H, D, preset)
    for s in (o, g):
        s.init()
        s.add_solid(50, 8, 9, 12, 7, 6)
    o.step(6); g.step(6)
    if rows:
        st = g.stats()
        assert st["fused_cells"] > 0
    o.step(4); g.step(4)                     # a second call straight behind the first: no fields in between
    for s in (o, g):
        s.set_params(inlet_velocity=0.05)    # the fields of the last step were made with the OLD parameters
    same(o, g, f"preset {preset}: two even calls, then the fields")
    o.step(8); g.step(8)                     # (this call writes rho/u as it goes: the caller read them before)
    same(o, g, f"preset {preset}: even call after a read")
    o.step(4); g.step(4)
    o.step(6); g.step(6)
    for s in (o, g):
        s.add_solid(90, 0, 4, 10, 9, 9)      # settles rho/u of the last step with the old mask first
    np.testing.assert_array_equal(g.density(), o.density())
    np.testing.assert_array_equal(g.velocity(), o.velocity())
    o.step(2); g.step(2)
    same(o, g, f"preset {preset}: after the mask change")
    o.step(10); g.step(10)
    g.set_field_window(3, 5)                 # drops the fields; asked for again they come through the same path
    np.testing.assert_array_equal(g.density(), np.asarray(o.density()).reshape(D, H, W)[3:8].ravel())


@pytest.mark.parametrize("preset", [0, 1])
@pytest.mark.parametrize("where", ["tail", "behind", "side"])
def test_where_the_rim_runs(oracle, lbm, preset, where, monkeypatch):
    """the rim's first step in the tail of the fused launch (k_fused2.h F2Rim: the workgroups behind the block list),
    both rim launches behind the fused kernel, or both on a side stream beside it: the same cells either way"""
    env = {"tail": {"LBM_FUSED_TAIL_PCT": "0"},
           "behind": {"LBM_FUSED_TAIL_RIM": "0", "LBM_FUSED_OVERLAP": "0"},
           "side": {"LBM_FUSED_TAIL_RIM": "0", "LBM_FUSED_OVERLAP_PCT": "0"}}[where]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    W, H, D = 200, 41, 38
    o = oracle.Solver(W, H, D, preset)
    g = lbm.Solver(W, H, D, preset)
    for s in (o, g):
        s.init()
        s.add_solid(60, 0, 10, 22, 15, 12)
        s.add_solid(W - 2, 20, 5, 2, 6, 9)
    for n in (7, 4, 12):
        o.step(n); g.step(n)
        same(o, g, f"preset {preset}, rim {where}, after {n} steps")
    fused_in_use(lbm, g)


@pytest.mark.parametrize("preset", [0, 1])
def test_rim_changes_place_inside_a_batch(oracle, lbm, preset):
    """long batches on a small domain with obstacles: while the library times the two forms of the first-step loads (its
    first 46 fused launches) the rim runs on the side stream, afterwards its first step rides in the fused launch -- the
    switch happens between two pairs of ONE lbm_step() call and must not let a launch start before the rim of the pair
    before it has ended"""
    W, H, D = 136, 40, 36
    o = oracle.Solver(W, H, D, preset)
    g = lbm.Solver(W, H, D, preset)
    for s in (o, g):
        s.init()
        s.add_solid(40, 0, 8, 20, 16, 14)
        s.add_solid(W - 2, 18, 5, 2, 6, 9)
    o.step(100); g.step(100)
    same(o, g, f"preset {preset}, 100 steps")
    o.step(31); g.step(31)
    same(o, g, f"preset {preset}, 131 steps")
    fused_in_use(lbm, g)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("zsplit", [0, 1, 3])
def test_fused_variants_with_obstacles(oracle, lbm, variant, zsplit, monkeypatch):
    monkeypatch.setenv("LBM_FUSED_VARIANT", variant)
    monkeypatch.setenv("LBM_FUSED_ZSPLIT", str(zsplit))
    W, H, D = 200, 40, 36
    for preset in (0, 1):
        o = oracle.Solver(W, H, D, preset)
        g = lbm.Solver(W, H, D, preset)
        for s in (o, g):
            s.init()
            s.add_solid(70, 0, 10, 20, 14, 9)        # on the floor, inside one tile column
            s.add_solid(120, 20, 20, 12, 6, 6)       # floating, straddles a tile border in x
            s.add_solid(W - 3, 30, 4, 3, 5, 5)       # in the outlet column
        o.step(9); g.step(9)
        same(o, g, f"variant {variant}, z split {zsplit}, preset {preset}")
        st = g.stats()
        # coarse tiles / a single z chunk around these obstacles may leave under a quarter of this small domain
        if fused_in_use(lbm, g, may_decline=not (variant == VARIANTS[0] and zsplit != 1)):
            assert 0 < st["fused_cells"] < W * H * D and st["rim_cells"] >= 0   # (walls go through the fused kernel too: the cavity has no rim)
            assert st["fused_cells"] + st["rim_cells"] <= W * H * D


@pytest.mark.parametrize("preset", [0, 1])
def test_fused_path_takes_a_domain_with_obstacles(oracle, lbm, preset):
    """a planner that silently declined (or barely used) a domain would still pass every parity test: on the
    row-interleaved layout at least half of this domain must go through the fused kernel, obstacles inside its
    blocks included, and equal the oracle"""
    import os
    if os.environ.get("LBM_LAYOUT") != str(lbm.LAYOUT_ROWS):
        pytest.skip("the fused path runs on the row-interleaved layout")
    W, H, D = 200, 60, 48
    o = oracle.Solver(W, H, D, preset)
    g = lbm.Solver(W, H, D, preset)
    for s in (o, g):
        s.init()
        s.add_solid(60, 0, 16, 30, 20, 14)
        s.add_solid(130, 30, 8, 10, 9, 30)
    o.step(7); g.step(7)
    same(o, g, f"preset {preset}")
    st = g.stats()
    assert st["fused_cells"] >= W * H * D // 2, st
    assert st["fused_cells"] + st["rim_cells"] <= W * H * D


@pytest.mark.parametrize("walls", ["1", "0"])
@pytest.mark.parametrize("preset", [0, 1])
def test_fused_blocks_on_the_walls(oracle, lbm, preset, walls, monkeypatch):
    """the WALL form of the kernel: blocks that lie on the domain's floor / ceiling rows and first / last layers take the
    reference's wall rules (floor bounce-back, ceiling reflect_y, side reflect_z; the cavity's walls and lid) in both
    of their steps.  The cavity then has no rim at all; the wind tunnel keeps the strips of its outlet tile column
    (quirk Q2).  LBM_FUSED_WALLS=0 is the earlier form with the blocks two cells off every wall."""
    import os
    if os.environ.get("LBM_LAYOUT") != str(lbm.LAYOUT_ROWS):
        pytest.skip("the fused path runs on the row-interleaved layout")
    monkeypatch.setenv("LBM_FUSED_WALLS", walls)
    W, H, D = 200, 37, 41
    o = oracle.Solver(W, H, D, preset)
    g = lbm.Solver(W, H, D, preset)
    for s in (o, g):
        s.init()
        s.set_params(inlet_velocity=0.07, omega_shear=1.75, omega_bulk=1.6)
        s.add_solid(40, H - 6, 0, 9, 6, 7)          # in the corner of ceiling and first layer
        s.add_solid(90, 10, D - 4, 12, 8, 4)        # on the last layer
    for n in (5, 4, 1, 2):
        o.step(n); g.step(n)
        same(o, g, f"preset {preset}, walls {walls}, after a call of {n} steps")
    st = g.stats()
    cells = W * H * D
    if walls == "1":
        if preset == lbm.PRESET_CAVITY:
            assert st["rim_cells"] == 0, st
        else:
            assert st["rim_cells"] <= 64 * (4 * D + 4 * H), st      # two rows / layers at every wall of one tile column
        assert st["fused_cells"] + st["rim_cells"] >= cells - int(o.solid_mask().sum()), st
    else:
        assert st["rim_cells"] >= W * 2 * (H + D), st


@pytest.mark.parametrize("form", ["0", "1", None])
@pytest.mark.parametrize("preset", [0, 1])
def test_fused_first_step_load_forms(oracle, lbm, preset, form, monkeypatch):
    """the plain blocks' first-step loads come in two forms (k_fused2.h WIDE: 16-byte loads of whole rows, the halo-column
    wave reading the row waves' staging areas; or one dword per lane and a loading halo-column wave).  Forced either way,
    or timed and chosen by the library (LBM_FUSED_WIDE unset), every cell equals the oracle; W ragged, x tiles at both
    ends of the domain and inside, obstacles (whose blocks keep the narrow form) beside plain blocks"""
    import os
    if os.environ.get("LBM_LAYOUT") != str(lbm.LAYOUT_ROWS):
        pytest.skip("the fused path runs on the row-interleaved layout")
    if form is None:
        monkeypatch.delenv("LBM_FUSED_WIDE", raising=False)
    else:
        monkeypatch.setenv("LBM_FUSED_WIDE", form)
    W, H, D = 270, 40, 30
    o = oracle.Solver(W, H, D, preset)
    g = lbm.Solver(W, H, D, preset)
    for s in (o, g):
        s.init()
        s.set_params(inlet_velocity=0.07, omega_shear=1.75, omega_bulk=1.6)
        s.add_solid(150, 12, 8, 10, 9, 11)
    for n in (7, 4, 95):         # 3 + 1 + 47 pairs: the library times the two forms over the first 46 launches of a plan
        o.step(n); g.step(n)
        same(o, g, f"preset {preset}, LBM_FUSED_WIDE {form}, after a call of {n} steps")
        if form is None and n == 4:
            assert g.stats()["fused_wide_loads"] == -1      # still alternating
    st = g.stats()
    assert st["fused_cells"] > 0
    if form is None:
        assert st["fused_wide_loads"] in (0, 1) and min(st["fused_tuned_ms"]) > 0, st
    else:
        assert st["fused_wide_loads"] == int(form) and max(st["fused_tuned_ms"]) == 0, st


def test_fused_cube_64(oracle, lbm):
    """BASELINE.json configs[0]: 64^3 channel around assets/cube.obj, 100 steps in calls of mixed length"""
    o = oracle.Solver(64, 64, 64)
    g = lbm.Solver(64, 64, 64)
    for s in (o, g):
        s.init(); s.load_mesh(CUBE_OBJ)
    for n in (10, 1, 7, 2, 80):
        o.step(n); g.step(n)
        same(o, g, f"after a call of {n}")
    assert abs(g.density().astype(np.float64).sum() - 274423.092220902) < 1e-6   # SURVEY.md Appendix A.2, N = 100
    fused_in_use(lbm, g)


def test_fused_mask_params_reset(oracle, lbm):
    """eligibility and rim lists follow the mask; parameters are read per call; reset keeps the mask"""
    W, H, D = 130, 30, 24
    o = oracle.Solver(W, H, D)
    g = lbm.Solver(W, H, D)
    o.init(); g.init()
    o.step(6); g.step(6)
    c0 = g.stats()["fused_cells"]
    for s in (o, g):
        s.add_solid(40, 8, 6, 30, 9, 9)
        s.set_params(inlet_velocity=0.05, omega_shear=1.7, omega_bulk=1.6)
    o.step(7); g.step(7)
    same(o, g, "after add_solid + set_params")
    if fused_in_use(lbm, g, may_decline=True):          # the box leaves little of this small domain to 12-row tiles
        assert g.stats()["fused_cells"] < c0
    m = o.solid_mask().reshape(D, H, W).copy()
    m[:, 8:17, 40:70] = 0                               # the box turns fluid again (frozen populations thaw)
    m[10:14, 20:25, 100:110] = 255
    o.set_solid(m); g.set_solid(m)
    o.step(5); g.step(5)
    same(o, g, "after set_solid")
    o.init(); g.init()
    o.step(4); g.step(4)
    same(o, g, "after reset")
    f = o.populations() * np.float32(1.01)
    o.set_populations(f); g.set_populations(f)
    o.step(5); g.step(5)
    same(o, g, "after a population import")


def test_fused_fields_on_demand_and_fused_moments(oracle, lbm):
    """the last step of a call is a single step: rho/u on demand, rho/u fused into it, curl, field window"""
    W, H, D = 72, 26, 20
    o = oracle.Solver(W, H, D)
    g = lbm.Solver(W, H, D)
    for s in (o, g):
        s.init(); s.add_solid(20, 0, 4, 9, 11, 9)
    for n in (5, 6, 3):                                 # the second and third call write rho/u in their last step
        o.step(n); g.step(n)
        np.testing.assert_array_equal(g.density(), o.density())
        np.testing.assert_array_equal(g.velocity(), o.velocity())
    o.compute_curl()
    np.testing.assert_array_equal(g.curl_field(), o.curl_field())
    g.set_field_window(6, 4)
    o.step(8); g.step(8)
    np.testing.assert_array_equal(g.density().reshape(4, -1), o.density().reshape(D, -1)[6:10])


@pytest.mark.parametrize("preset_name,wide", [("cavity", None), ("cavity", "1"), ("wind_tunnel", "1")])
def test_fused_equals_single_steps_256(lbm, preset_name, wide, monkeypatch):
    """BASELINE.json configs[1] at full size (256^3; the cavity as named and the reference's wind tunnel the bench
    headlines): 21 steps fused against 21 single steps, with the form of the first-step loads left to the library
    (narrow for a run this short) and with the wide form the headline settles on"""
    if wide is not None:
        monkeypatch.setenv("LBM_FUSED_WIDE", wide)
    preset = lbm.PRESET_CAVITY if preset_name == "cavity" else lbm.PRESET_WIND_TUNNEL
    a = lbm.Solver(256, 256, 256, preset, two_step=lbm.TWO_STEP_OFF)
    b = lbm.Solver(256, 256, 256, preset)
    for s in (a, b):
        s.init()
        if preset_name == "cavity":
            s.set_params(0.08, 1.8, 1.9)
        s.step(21)
    assert a.stats()["fused_cells"] == 0
    if fused_in_use(lbm, b):
        assert b.stats()["fused_cells"] >= 256 * 252 * 252   # (more where the planner lets blocks lie on the walls)
    np.testing.assert_array_equal(a.density(), b.density())
    np.testing.assert_array_equal(a.velocity(), b.velocity())
    fa, fb = a.populations(), b.populations()
    assert np.array_equal(fa, fb)


def test_fused_equals_single_steps_car(lbm, monkeypatch):
    """BASELINE.json configs[2] at full size (512 x 256 x 256 wind tunnel + the voxelised car stand-in): 21 + 12 steps fused
    -- blocks running through the car, the rim's first step in the tail of the fused launch (rim 6 % of the cells), an
    even call ending on a pair -- against single steps"""
    import os
    # the form of the first-step loads this domain settles on; left to the library it is still being timed over these
    # few launches, and the rim keeps out of the fused launch while it is
    monkeypatch.setenv("LBM_FUSED_WIDE", "0")
    car = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "car_like.obj")
    a = lbm.Solver(512, 256, 256, lbm.PRESET_WIND_TUNNEL, two_step=lbm.TWO_STEP_OFF)
    b = lbm.Solver(512, 256, 256, lbm.PRESET_WIND_TUNNEL)
    for s in (a, b):
        s.init()
        s.load_mesh(car)
        s.step(21)
        s.step(12)
    assert a.stats()["fused_cells"] == 0
    if fused_in_use(lbm, b):
        st = b.stats()
        assert st["fused_cells"] > 512 * 256 * 256 // 2 and st["fused_tail_cells"] > 0, st
    np.testing.assert_array_equal(a.density(), b.density())
    np.testing.assert_array_equal(a.velocity(), b.velocity())
    assert np.array_equal(a.populations(), b.populations())
    a.close(); b.close()


def test_fused_blocks_hold_obstacles(oracle, lbm, monkeypatch):
    """blocks run THROUGH obstacles (the kernel bounces back from solid cells itself, solid cells are never
    stored): a thin plate, a staircase, a box touching the floor, single cells; with LBM_FUSED_OBST=0 the
    blocks keep clear of them as in the first form of the path.  Both must equal the oracle bit for bit."""
    W, H, D = 200, 52, 40
    covered = {}
    for obst in ("1", "0"):
        monkeypatch.setenv("LBM_FUSED_OBST", obst)
        for preset in (0, 1):
            o = oracle.Solver(W, H, D, preset)
            g = lbm.Solver(W, H, D, preset)
            for s in (o, g):
                s.init()
                s.add_solid(60, 10, 8, 1, 30, 20)          # plate one cell thick across the flow
                s.add_solid(100, 0, 12, 30, 9, 14)         # box on the floor over a tile border in x
                for i in range(8):                          # staircase: diagonal neighbours only
                    s.add_solid(150 + i, 20 + i, 15 + i, 1, 1, 1)
                s.add_solid(63, 25, 20, 1, 1, 1)            # single cell at a tile's last column / next tile's first
                s.add_solid(64, 27, 22, 1, 1, 1)
                s.add_solid(0, 30, 10, 3, 6, 5)             # in the inlet column / on the cavity's x = 0 wall
                s.add_solid(W - 2, 14, 25, 2, 5, 4)         # in the outlet column / on the x = W-1 wall
            o.step(11); g.step(11)
            same(o, g, f"LBM_FUSED_OBST={obst}, preset {preset}")
            if fused_in_use(lbm, g, may_decline=obst == "0"):
                covered[(obst, preset)] = g.stats()["fused_cells"]
            o.step(4); g.step(4)                            # again from the state the fused path left
            same(o, g, f"LBM_FUSED_OBST={obst}, preset {preset}, second call")
    for preset in (0, 1):
        if ("1", preset) in covered:
            assert covered[("1", preset)] > covered.get(("0", preset), 0)


def test_fused_equals_single_steps_car_512(lbm):
    """BASELINE.json configs[2] at full size (512x256x256 wind tunnel around the voxelised car stand-in):
    13 steps with blocks through the car against 13 single steps"""
    import os
    car = os.path.join(os.path.dirname(__file__), "golden", "car_like.obj")
    a = lbm.Solver(512, 256, 256, two_step=lbm.TWO_STEP_OFF)
    b = lbm.Solver(512, 256, 256)
    for s in (a, b):
        s.init(); s.load_mesh(car); s.step(13)
    if fused_in_use(lbm, b):
        assert b.stats()["fused_cells"] > 0.6 * 512 * 256 * 256       # cells the kernel updates (solid cells inside blocks not counted)
    np.testing.assert_array_equal(a.density(), b.density())
    np.testing.assert_array_equal(a.velocity(), b.velocity())
    assert np.array_equal(a.populations(), b.populations())


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_fused_random_masks_equal_single_steps(lbm, seed):
    """random masks (scattered single cells at three densities plus random boxes: every tile has solids in
    reach) on both presets: the two-steps-per-pass path against the one-step kernel, which the oracle tests pin"""
    rng = np.random.default_rng(seed)
    W, H, D = 192 + 8 * seed, 40 + seed, 30 + 2 * seed
    m = (rng.random((D, H, W)) < (0.002, 0.02, 0.1, 0.3)[seed - 1]).astype(np.uint8) * 255
    for _ in range(6):
        x, y, z = rng.integers(0, W - 8), rng.integers(0, H - 6), rng.integers(0, D - 6)
        m[z:z + rng.integers(1, 6), y:y + rng.integers(1, 6), x:x + rng.integers(1, 40)] = 255
    for preset in (0, 1):
        a = lbm.Solver(W, H, D, preset, two_step=lbm.TWO_STEP_OFF)
        b = lbm.Solver(W, H, D, preset)
        for s in (a, b):
            s.init(); s.set_solid(m)
            if preset == 1:
                s.set_params(0.05, 1.6, 1.7)
        for n in (9, 4):
            a.step(n); b.step(n)
            np.testing.assert_array_equal(a.density(), b.density())
            np.testing.assert_array_equal(a.velocity(), b.velocity())
            assert np.array_equal(a.populations(), b.populations())
        assert a.stats()["fused_cells"] == 0
        fused_in_use(lbm, b, may_decline=seed == 4)
