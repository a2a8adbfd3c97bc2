"""fused_bench.py -- two-steps-per-pass variants against the one-step kernel (tuning probe, not product).
usage: python tools/fused_bench.py [W H D] [--preset cavity|tunnel] [--obj car]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

args = [a for a in sys.argv[1:] if not a.startswith("--")]
W, H, D = (int(a) for a in args[:3]) if len(args) >= 3 else (256, 256, 256)
preset_name = "tunnel" if "--tunnel" in sys.argv else "cavity"
car = "--car" in sys.argv
variants = os.environ.get("FB_VARIANTS", "off;9,1,2;12,1,1;13,1,1;9,0,2;12,0,2").split(";")
zsplits = [int(z) for z in os.environ.get("FB_ZSPLITS", "0").split(",")]
steps = int(os.environ.get("FB_STEPS", "201"))

from cfd_amd import lbm
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in variants:
    for zs in (zsplits if v != "off" else [0]):
        if v == "off":
            os.environ["LBM_FUSED"] = "0"
        else:
            os.environ["LBM_FUSED"] = "1"
            os.environ["LBM_FUSED_VARIANT"] = v
            os.environ["LBM_FUSED_ZSPLIT"] = str(zs)
        s = lbm.Solver(W, H, D, lbm.PRESET_CAVITY if preset_name == "cavity" else lbm.PRESET_WIND_TUNNEL)
        s.init()
        if preset_name == "cavity":
            s.set_params(0.08, 1.8, 1.9)
        if car:
            s.load_mesh(os.path.join(ROOT, "tests", "golden", "car_like.obj"))
        s.step(21); s.sync()
        best = None
        for rep in range(3):
            s.timer_begin()
            t0 = time.perf_counter()
            s.step(steps)
            ms = s.timer_end()
            dt = time.perf_counter() - t0
            st = s.stats()
            if best is None or ms < best[0]:
                best = (ms, dt, st)
        ms, dt, st = best
        mlups = W * H * D * steps / (ms * 1e-3) / 1e6
        line = f"{preset_name} {W}x{H}x{D}{' car' if car else ''} variant {v:8s} zsplit {zs}: {mlups:8.0f} MLUPS ({ms / steps * 1e3:7.1f} us/step)"
        if st["last_timed_fused_launches"]:
            fl, fms = st["last_timed_fused_launches"], st["last_timed_fused_ms"]
            line += (f" | fused {st['fused_cells'] / (W * H * D) * 100:5.1f} % of cells, {fms / fl * 1e3:7.1f} us per launch = "
                     f"{2 * st['fused_cells'] / (fms / fl * 1e-3) / 1e6:8.0f} MLUPS inside, rim+single {(ms - fms) / steps * 1e3:6.1f} us/step")
        rho = s.density()
        line += f" | finite {bool(np.isfinite(rho).all())}"
        print(line, flush=True)
        s.close()
