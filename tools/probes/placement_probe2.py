"""Is the bimodal step time a property of the solver instance (its allocation) or of the moment?
Three live instances timed in turn, several rounds (ad-hoc probe).  usage: placement_probe2.py W H D preset"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfd_amd import lbm
W, H, D = (int(a) for a in sys.argv[1:4])
preset = int(sys.argv[4])
n = W * H * D
os.environ["LBM_DEBUG_PTRS"] = "1"
inst = []
for i in range(3):
    s = lbm.Solver(W, H, D, preset=preset)
    s.set_field_window(0, 0)
    s.init(); s.step(4); s.sync()
    inst.append(s)
for rnd in range(4):
    out = []
    for s in inst:
        reps = 10
        s.timer_begin(); s.step(reps); ms = s.timer_end() / reps
        out.append(f"{n*152/ms/1e9:.3f}")
    print(f"{W}x{H}x{D} round {rnd}: " + "  ".join(out) + " TB/s", flush=True)
    if rnd == 1:
        for s in inst:
            s.init()
        print("re-init", flush=True)
