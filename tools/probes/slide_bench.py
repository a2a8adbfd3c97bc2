"""Sliding in-place scheme: step time against chunk size, row padding and read/write skew, up / down
sweeps apart (ad-hoc probe).  chunk 0 = A/B in one launch, chunk < 0 = A/B in launches of -chunk layers.
usage: slide_bench.py W H D preset chunk[,chunk..] [row_pad[,row_pad..] [skew_bytes[,..]]]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfd_amd import lbm
W, H, D = (int(a) for a in sys.argv[1:4])
preset = int(sys.argv[4])
chunks = [int(a) for a in sys.argv[5].split(",")]
pads = [int(a) for a in sys.argv[6].split(",")] if len(sys.argv) > 6 else [0]
skews = [int(a) for a in sys.argv[7].split(",")] if len(sys.argv) > 7 else [0]
n = W * H * D
for pad in pads:
    os.environ["LBM_ROW_PAD"] = str(pad)
    for skew in skews:
        os.environ["LBM_SLIDE_SKEW"] = str(skew)
        for chunk in chunks:
            scheme = lbm.SCHEME_AB if chunk <= 0 else lbm.SCHEME_SLIDE
            os.environ["LBM_AB_CHUNK"] = str(-chunk if chunk < 0 else 0)
            s = lbm.Solver(W, H, D, preset=preset, scheme=scheme, slide_chunk=max(chunk, 0))
            s.set_field_window(0, 0)
            s.init(); s.step(6); s.sync()
            t = [0.0, 0.0]
            reps = 10
            for i in range(2 * reps):
                s.timer_begin(); s.step(1); t[i & 1] += s.timer_end()
            a, b = t[0] / reps, t[1] / reps
            print(f"{W}x{H}x{D} preset={preset} pad={pad} skew={skew} chunk={chunk if chunk > 0 else 'A/B' + (str(chunk) if chunk else '')}: "
                  f"up {a*1e3:.1f} us {n*152/a/1e9:.3f} TB/s, down {b*1e3:.1f} us {n*152/b/1e9:.3f} TB/s", flush=True)
            s.close()
