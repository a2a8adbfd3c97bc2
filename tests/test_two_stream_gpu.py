"""SlabSolver's real per-step schedule on ONE GPU: a one-rank ring whose messages are device copies
(cfd_amd.slab.LoopbackP2P), so the two-stream ordering -- boundary layers and exchange on a side stream
beside the interior update -- runs exactly as it does under RCCL.  A one-slab ring of the wind-tunnel
preset is the undecomposed domain, so the result must equal the plain A/B solver bit for bit.

Covers what the round-1 review found untested:
  * the sliding scheme on two streams with the interior cut into many chunks (the interior's later
    chunks overwrite the layers the boundary launches read: lbm_hip.hip slide_layers());
  * the schedule with more live streams than hardware queues (GPU_MAX_HW_QUEUES = 4: streams that share a
    queue serialise, and a wait for work queued BEHIND the waiter on the same queue would never return).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def reference_run(lbm, W, H, D, steps, rects):
    g = lbm.Solver(W, H, D)
    g.init()
    for r in rects:
        g.add_solid(*r)
    g.step(steps)
    out = g.density(), g.velocity(), g.populations(), g.solid_mask() == 0
    g.close()
    return out


def ring_run(lbm, W, H, D, steps, rects, scheme, two_streams, chunk=0, monkeypatch=None):
    import torch
    from cfd_amd import slab
    if chunk:
        monkeypatch.setenv("LBM_SLIDE_CHUNK", str(chunk))
        monkeypatch.setenv("LBM_AB_CHUNK", str(chunk))
    e = slab.HipSlabEngine(W, H, D, 0, D, lbm.PRESET_WIND_TUNNEL, scheme=scheme, two_streams=two_streams)
    drv = slab.SlabSolver(e, lbm.PRESET_WIND_TUNNEL, 1, 0, slab.LoopbackP2P())
    s = e.solver
    s.init()
    for r in rects:
        s.add_solid(*r)
    drv.exchange_mask()
    drv.exchange_populations()
    drv.step(steps)
    torch.cuda.synchronize()
    out = s.density(), s.velocity(), s.populations()
    s.close()
    return out


RECTS = [(20, 0, 3, 9, 6, 30), (70, 3, 10, 2, 4, 9)]      # the second one sits in the outlet column


@pytest.mark.timeout(120)
@pytest.mark.parametrize("chunk", [1, 2, 5])
def test_slide_two_streams_many_chunks(lbm, chunk, monkeypatch):
    """LBM_SCHEME_SLIDE split step on two streams, D much larger than the chunk, odd and even steps"""
    W, H, D = 72, 12, 40
    for steps in (4, 7):
        rho, u, f, _ = reference_run(lbm, W, H, D, steps, RECTS)
        r2, u2, f2 = ring_run(lbm, W, H, D, steps, RECTS, lbm.SCHEME_SLIDE, True, chunk, monkeypatch)
        np.testing.assert_array_equal(r2, rho)
        np.testing.assert_array_equal(u2, u)
        np.testing.assert_array_equal(f2, f)       # solid cells included (the sliding scheme carries them)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("scheme", [0, 1, 2])
@pytest.mark.parametrize("two_streams", [False, True])
def test_schedule_with_more_streams_than_queues(lbm, scheme, two_streams, monkeypatch):
    """twelve busy streams alive beside the solver's: its streams share hardware queues with them"""
    import torch
    W, H, D = 72, 12, 24
    rho, u, f, fluid = reference_run(lbm, W, H, D, 6, RECTS)
    others = [torch.cuda.Stream() for _ in range(12)]
    junk = []
    for st in others:
        with torch.cuda.stream(st):
            junk.append(torch.zeros(1 << 20, device="cuda").add_(1.0))
    r2, u2, f2 = ring_run(lbm, W, H, D, 6, RECTS, scheme, two_streams, 3, monkeypatch)
    np.testing.assert_array_equal(r2, rho)
    np.testing.assert_array_equal(u2, u)
    np.testing.assert_array_equal(f2[:, fluid], f[:, fluid])
    torch.cuda.synchronize()
    assert all(float(j[0]) == 1.0 for j in junk)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("scheme,two_streams", [(0, True), (0, False), (2, True)])
def test_ring_of_one_over_the_peer_copy_transport(lbm, scheme, two_streams):
    """cfd_amd/slab.py IpcCopyP2P on a one-rank ring: the copies stay inside the process, the flags and their order are
    those of an N-rank run (tests/test_two_ranks_one_gpu.py has the processes).  Pairs and single split steps."""
    import torch
    from cfd_amd import slab
    W, H, D = 136, 30, 28
    rho, u, f, fluid = reference_run(lbm, W, H, D, 9, RECTS)
    e = slab.HipSlabEngine(W, H, D, 0, D, lbm.PRESET_WIND_TUNNEL, scheme=scheme, two_streams=two_streams)
    t = slab.IpcCopyP2P(slab.LoopbackP2P(), None, 0, 1, timeout_ms=5000)
    drv = slab.SlabSolver(e, lbm.PRESET_WIND_TUNNEL, 1, 0, t)
    s = e.solver
    try:
        assert t.links is not None, t.error
        assert t.verify(), t.error
        s.init()
        for r in RECTS:
            s.add_solid(*r)
        drv.exchange_mask()
        drv.exchange_populations()
        for n in (5, 1, 3):
            drv.step(n)
        torch.cuda.synchronize()
        assert not t.timed_out()
        np.testing.assert_array_equal(s.density(), rho)
        np.testing.assert_array_equal(s.velocity(), u)
        np.testing.assert_array_equal(s.populations()[:, fluid], f[:, fluid])
    finally:
        t.close()
        s.close()


@pytest.mark.timeout(60)
def test_flag_waits_give_up(lbm):
    """include/lbm_abi.h lbm_flag_await / lbm_flag_sync: a wait nobody answers ends after its time limit, raises the
    rank's timed-out word and lets the stream go on; waits behind it return at once; sequence numbers wrap"""
    import ctypes as C
    import time
    import torch
    lib = lbm.load_library()
    blk = C.c_void_p()
    assert lib.lbm_ipc_alloc(4 * 16 * 4, 1, C.byref(blk), None) == 0, lib.lbm_last_error(None)
    flag, other, timed_out = blk.value, blk.value + 64, blk.value + 128
    stream = torch.cuda.Stream()
    st = stream.cuda_stream

    def peek(p):
        v = C.c_uint32(0)
        assert lib.lbm_flag_peek(p, C.byref(v)) == 0
        return v.value
    try:
        # answered waits: posted before, posted with a larger number, and across the wrap of the 32-bit counter
        assert lib.lbm_flag_post(st, flag, 7) == 0
        assert lib.lbm_flag_await(st, flag, 7, 2000, timed_out) == 0
        assert lib.lbm_flag_await(st, flag, 5, 2000, timed_out) == 0
        assert lib.lbm_flag_post(st, flag, 0xfffffffe) == 0
        assert lib.lbm_flag_post(st, flag, 2) == 0            # = 0xfffffffe + 4
        assert lib.lbm_flag_await(st, flag, 0xffffffff, 2000, timed_out) == 0
        pf, pv = (C.c_void_p * 2)(other, flag), (C.c_uint32 * 2)(11, 3)
        wf, wv = (C.c_void_p * 2)(flag, other), (C.c_uint32 * 2)(3, 11)
        assert lib.lbm_flag_sync(st, pf, pv, 2, wf, wv, 2, 2000, timed_out) == 0
        stream.synchronize()
        assert peek(timed_out) == 0 and peek(flag) == 3 and peek(other) == 11
        # nobody posts 100
        t0 = time.perf_counter()
        assert lib.lbm_flag_await(st, flag, 100, 300, timed_out) == 0
        stream.synchronize()
        dt = time.perf_counter() - t0
        assert peek(timed_out) == 1 and 0.25 < dt < 5.0, dt
        t0 = time.perf_counter()
        for _ in range(20):
            assert lib.lbm_flag_await(st, flag, 100, 300, timed_out) == 0
        wf1, wv1 = (C.c_void_p * 1)(flag), (C.c_uint32 * 1)(100)
        assert lib.lbm_flag_sync(st, None, None, 0, wf1, wv1, 1, 300, timed_out) == 0
        stream.synchronize()
        assert time.perf_counter() - t0 < 1.0          # (21 x 0.3 s if they waited)
        # bad arguments are refused on the host
        assert lib.lbm_flag_await(st, flag, 1, 0, timed_out) != 0
        assert lib.lbm_flag_sync(st, pf, pv, 5, None, None, 0, 300, timed_out) != 0
        assert lib.lbm_flag_post(st, None, 1) != 0
    finally:
        torch.cuda.synchronize()
        assert lib.lbm_ipc_free(blk) == 0


def test_message_buffers_are_checked(lbm):
    """the library takes raw pointers: the Python engine refuses buffers the boundary kernel would overrun"""
    import torch
    from cfd_amd import slab
    e = slab.HipSlabEngine(64, 8, 12, 0, 6, lbm.PRESET_WIND_TUNNEL)
    e.solver.init()
    short = torch.zeros(e.message_floats() - 1, dtype=torch.float32, device=e.device)
    wrong = torch.zeros(e.message_floats(), dtype=torch.float64, device=e.device)
    for bad in (short, wrong, torch.zeros(e.message_floats())):
        with pytest.raises(ValueError):
            e.step_boundary(bad, None)
        with pytest.raises(ValueError):
            e.halo_unpack_both(None, bad)
    ok = e.new_message()
    e.step_boundary(ok, None)
    e.step_interior()
    e.step_finish()
    e.solver.close()
