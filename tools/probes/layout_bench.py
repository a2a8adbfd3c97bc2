"""Step time of both layouts / both schemes at 256^3 (ad-hoc probe; LBM_LAYOUT / LBM_TUNE from env)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfd_amd import lbm

def run(W, H, D, preset, scheme, layout, steps=100, warm=20):
    s = lbm.Solver(W, H, D, preset=preset, scheme=scheme, layout=layout)
    s.init(); s.step(warm); s.sync()
    best = 1e9
    for _ in range(3):
        s.timer_begin(); s.step(steps); ms = s.timer_end()
        best = min(best, ms / steps)
    n = W * H * D
    mlups = n / best / 1e3
    print(f"{'AA' if scheme else 'AB'} layout={layout} {W}x{H}x{D} preset={preset} tune={os.environ.get('LBM_TUNE','def')}: {best*1e3:.1f} us/step {mlups:.0f} MLUPS {mlups*152e6/1e12:.3f} TB/s", flush=True)
    s.close()

if __name__ == "__main__":
    dims = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 256, 256)
    for layout in (2, 1):
        for scheme in (0, 1):
            for preset in (1, 0):
                run(*dims, preset, scheme, layout)
