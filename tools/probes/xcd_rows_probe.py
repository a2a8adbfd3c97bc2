"""Whole rows of workgroups per XCD (ZORDER_XCD, k_step.h) against the plain launch order, on ONE
solver instance.  usage: xcd_rows_probe.py W H D preset chunk [ab|aa]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfd_amd import lbm
W, H, D = (int(a) for a in sys.argv[1:4])
preset = int(sys.argv[4])
chunk = sys.argv[5]
scheme = {"ab": lbm.SCHEME_AB, "aa": lbm.SCHEME_AA}[sys.argv[6]] if len(sys.argv) > 6 else lbm.SCHEME_AB
n = W * H * D
os.environ["LBM_PROBE"] = "1"
os.environ["LBM_AB_CHUNK"] = chunk
for inst in range(2):
    s = lbm.Solver(W, H, D, preset=preset, scheme=scheme)
    s.set_field_window(0, 0)
    s.init(); s.step(6); s.sync()
    out = []
    for xcd in (0, 1, 0, 1, 0, 1):
        os.environ["LBM_XCD_ROWS"] = str(xcd)
        s.step(2)
        s.timer_begin(); s.step(10); ms = s.timer_end() / 10
        out.append(f"{'xcd' if xcd else 'std'}:{n*152/ms/1e9:.3f}")
    print(f"{W}x{H}x{D} preset={preset} {sys.argv[6] if len(sys.argv) > 6 else 'ab'} chunk={chunk} instance {inst}: " + " ".join(out), flush=True)
    s.close()
