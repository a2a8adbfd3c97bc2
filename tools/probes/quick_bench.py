"""Ad-hoc timing probe used while tuning (not the driver's bench.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfd_amd import lbm

def run(W, H, D, preset, steps=100, warm=20, skew=-1, label=""):
    label = label or ("tune=" + os.environ.get("LBM_TUNE", "0"))
    s = lbm.Solver(W, H, D, preset=preset, plane_skew_bytes=skew)
    s.init(); s.step(warm); s.sync()
    best = 1e9
    for _ in range(3):
        s.timer_begin(); s.step(steps); ms = s.timer_end()
        best = min(best, ms / steps)
    n = W * H * D
    mlups = n / best / 1e3
    print(f"{label or ''} {W}x{H}x{D} preset={preset} skew={skew}: {best*1e3:.1f} us/step  {mlups:.0f} MLUPS  {mlups*152e6/1e12:.3f} TB/s  ({mlups*152e6/8e12*100:.1f}% of 8 TB/s)", flush=True)
    s.close()

if __name__ == "__main__":
    for preset in (1, 0):
        run(256, 256, 256, preset)
    run(512, 256, 256, 0)
    run(512, 512, 512, 0, steps=30, warm=5)
    for skew in (0, 256, 1024, 4096, 4352, 8448, 65536+256):
        run(256, 256, 256, 0, skew=skew)
