"""A/B step time against layers per launch on ONE solver instance (the allocation, whose physical
placement moves the step time by +-3 %, stays the same).  usage: ab_chunk_probe.py W H D preset chunks [ab|aa]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfd_amd import lbm
W, H, D = (int(a) for a in sys.argv[1:4])
preset = int(sys.argv[4])
chunks = [int(a) for a in sys.argv[5].split(",")]
scheme = {"ab": lbm.SCHEME_AB, "aa": lbm.SCHEME_AA}[sys.argv[6]] if len(sys.argv) > 6 else lbm.SCHEME_AB
n = W * H * D
os.environ["LBM_PROBE"] = "1"
for inst in range(3):
    s = lbm.Solver(W, H, D, preset=preset, scheme=scheme)
    s.set_field_window(0, 0)
    s.init(); s.step(6); s.sync()
    out = []
    for chunk in chunks + chunks[::-1]:
        os.environ["LBM_AB_CHUNK"] = str(chunk)
        s.step(2)
        s.timer_begin(); s.step(10); ms = s.timer_end() / 10
        out.append(f"{chunk}:{n*152/ms/1e9:.3f}")
    print(f"{W}x{H}x{D} preset={preset} instance {inst}: " + " ".join(out), flush=True)
    s.close()
