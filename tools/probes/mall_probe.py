"""Does a domain that fits the 256 MiB Infinity Cache step faster?  (ad-hoc probe)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfd_amd import lbm

def run(W, H, D, preset, scheme=0, steps=200, warm=20):
    s = lbm.Solver(W, H, D, preset=preset, scheme=scheme)
    s.init(); s.step(warm); s.sync()
    best = 1e9
    for _ in range(3):
        s.timer_begin(); s.step(steps); ms = s.timer_end()
        best = min(best, ms / steps)
    n = W * H * D
    mlups = n / best / 1e3
    mb = n * 76 / 1e6
    print(f"{'AA' if scheme else 'AB'} {W}x{H}x{D} preset={preset} buf={mb:.0f}MB: {best*1e3:.1f} us/step {mlups:.0f} MLUPS {mlups*152e6/1e12:.3f} TB/s", flush=True)
    s.close()

if __name__ == "__main__":
    if len(sys.argv) > 1:  # quick: A/B only, few sizes, current LBM_TUNE
        for dims in ((256, 256, 8), (256, 256, 16), (256, 256, 24), (256, 256, 64)):
            run(*dims, 1, 0)
        sys.exit(0)
    for scheme in (0, 1):
        for dims in ((256, 256, 4), (256, 256, 8), (256, 256, 16), (256, 256, 24), (256, 256, 32), (256, 256, 48), (256, 256, 64), (256, 256, 128), (128, 128, 64), (512, 256, 8)):
            run(*dims, 1, scheme)
    run(256, 256, 16, 0, 0)
    run(256, 256, 16, 0, 1)
