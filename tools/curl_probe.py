"""Cost of the curl pass inside the frame loop (lbm_step_frames): frames with and without it.  usage: [W H D]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_amd import lbm
W, H, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 256, 256)
s = lbm.Solver(W, H, D)
s.init()
res = {}
for curl in (False, True, False, True):
    s.step_frames(20, curl=curl); s.sync()
    n = 200
    t0 = time.perf_counter(); s.step_frames(n, curl=curl); s.sync()
    res.setdefault(curl, []).append((time.perf_counter() - t0) / n * 1e6)
a, b = min(res[False]), min(res[True])
cells = W * H * D
print(f"{W}x{H}x{D}: frame {a:.1f} us ({cells*172/a/1e3:.0f} GB/s at 172 B), with curl {b:.1f} us ({cells*192/b/1e3:.0f} GB/s at 192 B = "
      f"{cells*192/b/1e3/8000:.3f} of peak), curl pass {b-a:.1f} us ({cells*20/(b-a)/1e3:.0f} GB/s at 20 B)", flush=True)
s.close()
