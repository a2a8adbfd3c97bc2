"""Does the step time depend on WHERE the two A/B population buffers lie?  Fresh solvers, default
two allocations against one allocation with a chosen distance between the buffers (ad-hoc probe).
usage: placement_probe.py W H D preset"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfd_amd import lbm
W, H, D = (int(a) for a in sys.argv[1:4])
preset = int(sys.argv[4])
n = W * H * D
os.environ["LBM_DEBUG_PTRS"] = "1"


def run(tag):
    s = lbm.Solver(W, H, D, preset=preset)
    s.set_field_window(0, 0)
    s.init(); s.step(6); s.sync()
    reps = 20
    s.timer_begin(); s.step(reps); ms = s.timer_end() / reps
    print(f"{W}x{H}x{D} preset={preset} {tag}: {ms*1e3:.1f} us {n*152/ms/1e9:.3f} TB/s", flush=True)
    s.close()


keep = []
for i in range(6):
    os.environ.pop("LBM_AB_SINGLE", None)
    run(f"two allocations #{i}")
    if i % 2:  # shift what the allocator hands out next
        import torch
        keep.append(torch.empty((i * 37 + 11) << 20, dtype=torch.uint8, device="cuda"))
for skew in (0, 128, 1024, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 3 << 20, 8 << 20, 0, 4096, 1 << 20):
    os.environ["LBM_AB_SINGLE"] = str(skew)
    run(f"one allocation, skew {skew}")
