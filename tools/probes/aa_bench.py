import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfd_amd import lbm
def run(W, H, D, preset, scheme, steps=100, warm=20, label=""):
    s = lbm.Solver(W, H, D, preset=preset, scheme=scheme)
    s.init(); s.step(warm); s.sync()
    best = 1e9
    for _ in range(3):
        s.timer_begin(); s.step(steps); best = min(best, s.timer_end() / steps)
    # per-phase timing: single steps
    ph = [1e9, 1e9]
    for i in range(20):
        s.timer_begin(); s.step(1); ph[i % 2] = min(ph[i % 2], s.timer_end())
    m = W * H * D / best / 1e3
    print(f"{label} {W}x{H}x{D} preset={preset} scheme={'AA' if scheme else 'AB'}: {best*1e3:.1f} us/step {m:.0f} MLUPS {m*152e6/1e12:.3f} TB/s  single-step phases {ph[0]*1e3:.0f}/{ph[1]*1e3:.0f} us", flush=True)
    s.close()
for tune in range(8):
    os.environ["LBM_TUNE"] = str(tune)
    run(256, 256, 256, 0, 1, label=f"tune={tune}")
