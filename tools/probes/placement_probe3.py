"""256^3 cavity, A/B: two allocations against both buffers in one allocation (LBM_AB_SINGLE), fresh
solvers in alternation (ad-hoc probe)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfd_amd import lbm
W, H, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 256, 256)
preset = int(sys.argv[4]) if len(sys.argv) > 4 else 1
n = W * H * D
res = {"two": [], "one": [], "one+2M": []}
for i in range(8):
    for tag, env in (("two", None), ("one", "0"), ("one+2M", str(2 << 20))):
        if env is None: os.environ.pop("LBM_AB_SINGLE", None)
        else: os.environ["LBM_AB_SINGLE"] = env
        s = lbm.Solver(W, H, D, preset=preset)
        s.set_field_window(0, 0)
        s.init(); s.step(10); s.sync()
        s.timer_begin(); s.step(40); ms = s.timer_end() / 40
        res[tag].append(n * 152 / ms / 1e9)
        s.close()
for tag, v in res.items():
    print(f"{W}x{H}x{D} {tag:7s}: " + " ".join(f"{x:.3f}" for x in v) + f"   mean {sum(v)/len(v):.3f}", flush=True)
