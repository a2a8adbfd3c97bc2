"""Even vs odd AA step time (ad-hoc probe)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfd_amd import lbm
dims = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 256, 256)
for preset in (1, 0):
    s = lbm.Solver(*dims, preset=preset, scheme=lbm.SCHEME_AA)
    s.set_field_window(0, 0)
    s.init(); s.step(10); s.sync()
    t = [0.0, 0.0]
    for i in range(40):
        s.timer_begin(); s.step(1); t[i & 1] += s.timer_end()
    n = dims[0] * dims[1] * dims[2]
    for ph in (0, 1):
        ms = t[ph] / 20
        print(f"AA {dims} preset={preset} tune={os.environ.get('LBM_TUNE','def')} {'even' if ph == 0 else 'odd '} step: {ms*1e3:.1f} us {n*152/ms/1e9:.3f} TB/s", flush=True)
    s.close()
