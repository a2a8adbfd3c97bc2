"""lbm_read_velocity_strided_noalloc (include/lbm_abi.h): the allocation-free twin of
lbm_read_velocity_strided, for callers that cannot call hipMalloc (graph capture, per-frame loops).
Its sampling kernel stages the samples in a fixed array inside the library's code object instead of a
hipMalloc'ed buffer.  Checked through the C ABI against the allocating entry point, bit for bit, on two
calls in a row (the second one smaller, so stale contents of the shared array would show), and against
the oracle's velocity field.  Collected last on purpose (pytest -x stops at the first failure)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_noalloc_equals_allocating_entry_point(oracle):
    from cfd_amd import lbm
    lbm.load_library()
    W, H, D = 96, 40, 24
    g = lbm.Solver(W, H, D)
    o = oracle.Solver(W, H, D)
    for s in (o, g):
        s.init(); s.add_solid(30, 0, 6, 11, 13, 9); s.step(9)
    # the call allocates nothing, rho/u included: before any field was read it must refuse
    with pytest.raises(lbm.LbmError) as e:
        g.velocity_strided(2, 2, 2, noalloc=True)
    assert e.value.code == -5
    u = o.velocity().reshape(D, H, W, 4)
    big = g.velocity_strided(1, 1, 1)                       # allocating entry point (also creates rho/u)
    np.testing.assert_array_equal(big, u)
    # call 1: every cell (92 160 samples); call 2, right behind it: a much smaller sample
    a1 = g.velocity_strided(1, 1, 1, noalloc=True)
    a2 = g.velocity_strided(5, 3, 4, noalloc=True)
    np.testing.assert_array_equal(a1, big)
    np.testing.assert_array_equal(a2, g.velocity_strided(5, 3, 4))
    np.testing.assert_array_equal(a2, u[::4, ::3, ::5])
    # after more steps, and through a field window
    o.step(4); g.step(4)
    u = o.velocity().reshape(D, H, W, 4)
    np.testing.assert_array_equal(g.velocity_strided(3, 2, 1, noalloc=True), u[::1, ::2, ::3])
    g.set_field_window(4, 10)                               # frees rho/u: the call must refuse again
    with pytest.raises(lbm.LbmError):
        g.velocity_strided(3, 2, 1, noalloc=True)
    np.testing.assert_array_equal(g.velocity_strided(3, 2, 3), u[4:14][::3, ::2, ::3])
    np.testing.assert_array_equal(g.velocity_strided(3, 2, 3, noalloc=True), u[4:14][::3, ::2, ::3])


def test_noalloc_rejects_oversize_and_bad_arguments():
    from cfd_amd import lbm
    L = lbm.load_library()
    g = lbm.Solver(128, 64, 64)                             # 524 288 cells > LBM_NOALLOC_MAX_SAMPLES (262 144)
    g.init(); g.step(1)
    g.velocity()                                            # field buffers exist
    with pytest.raises(lbm.LbmError) as e:
        g.velocity_strided(1, 1, 1, noalloc=True)
    assert e.value.code == -1 and b"exceed" in L.lbm_last_error(g._ctx)
    with pytest.raises(lbm.LbmError):
        g.velocity_strided(0, 1, 1, noalloc=True)
    full = g.velocity_strided(2, 1, 1)                      # exactly the maximum
    np.testing.assert_array_equal(g.velocity_strided(2, 1, 1, noalloc=True), full)
    # a second context on the same GPU shares the array (sequential calls: fine)
    h = lbm.Solver(32, 16, 8)
    h.init(); h.step(2); h.velocity()
    np.testing.assert_array_equal(h.velocity_strided(2, 2, 2, noalloc=True), h.velocity_strided(2, 2, 2))
    np.testing.assert_array_equal(g.velocity_strided(4, 4, 4, noalloc=True), g.velocity_strided(4, 4, 4))
