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
