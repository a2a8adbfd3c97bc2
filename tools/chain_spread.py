#!/usr/bin/env python3
"""How evenly the block chains of one k_fused2 launch end (lbm_fused_chain_stamps): per workgroup the time from the
launch's first start to its own end, as a share of the launch.

  python tools/chain_spread.py W H D [--preset cavity|wind_tunnel] [--car] [--slab Z0 DG] [--steps N]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_amd import lbm  # noqa: E402

CAR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "car_like.obj")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("size", type=int, nargs=3)
    ap.add_argument("--preset", default="wind_tunnel")
    ap.add_argument("--car", action="store_true")
    ap.add_argument("--slab", type=int, nargs=2, default=None, metavar=("Z0", "DG"))
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--blocks", default=None, help="file with the library's [fused-block] lines (LBM_FUSED_DEBUG=4, stderr of an earlier run of the same domain): durations by kind of block")
    a = ap.parse_args()
    W, H, D = a.size
    kw = dict(z0=a.slab[0], depth_global=a.slab[1]) if a.slab else {}
    s = lbm.Solver(W, H, D, preset=lbm.PRESET_CAVITY if a.preset == "cavity" else lbm.PRESET_WIND_TUNNEL, **kw)
    s.init()
    if a.preset == "cavity":
        s.set_params(0.08, 1.8, 1.9)
    if a.car:
        s.load_mesh(CAR)
    if not a.slab:
        s.step(a.steps)
    st = s.fused_chain_stamps()
    if st is None:
        print("the two-steps-per-pass path is not in use on this domain")
        return
    st = s.fused_chain_stamps()            # second launch: warm
    ran = st[:, 1] > 0
    t0 = st[ran, 0].min()
    start = (st[ran, 0] - t0) / 100.0      # us
    end = (st[ran, 1] - t0) / 100.0
    total = end.max()
    print(f"{W}x{H}x{D}: {int(ran.sum())} workgroups ran of {len(st)}; launch {total:.1f} us")
    print(f"  start: max {start.max():.1f} us, {int((start > 5).sum())} workgroups started later than 5 us")
    q = np.percentile(end, [0, 5, 25, 50, 75, 95, 100])
    print("  end (us): min %.1f  p5 %.1f  p25 %.1f  median %.1f  p75 %.1f  p95 %.1f  max %.1f" % tuple(q))
    print(f"  mean end / max end = {end.mean() / total:.4f} (1 = every CU busy to the end)")
    dur = end - start
    order = np.argsort(start)
    n = len(order)
    print("  workgroups in start order, in groups of 128: start (mean), duration (mean, min, max) in us")
    for i in range(0, n, 128):
        g = order[i:i + 128]
        print(f"    {i:5d}..{i + len(g) - 1:5d}: start {start[g].mean():8.1f}  duration {dur[g].mean():7.1f} [{dur[g].min():7.1f}, {dur[g].max():7.1f}]")
    if a.blocks:
        info = {}
        for line in open(a.blocks):
            if line.startswith("[fused-block]"):
                v = [int(t) for t in line.split()[1:]]
                info[v[0]] = v[1:]
        idx = np.arange(len(st))[ran]
        kinds = {}
        for j, wg in enumerate(idx):
            if wg not in info:
                continue
            bx, y0, n, zs, ln, clean = info[wg]
            kind = ("bottom " if y0 == 0 else "") + ("top " if y0 + n == H else "") + ("first " if zs == 1 else "") + ("last " if zs + ln == D + 1 else "")
            kinds.setdefault(f"bx {bx} " + (kind or "inner ") + f"rows {n} layers {ln} clean {clean}", []).append(dur[j])
        print("  duration by kind of block (us): count, mean, min, max, us per layer")
        for k in sorted(kinds):
            d = np.array(kinds[k])
            ln = int(k.split("layers ")[1].split()[0])
            print(f"    {k:48s} {len(d):5d}  {d.mean():7.1f} [{d.min():7.1f}, {d.max():7.1f}]  {d.mean() / (ln + 2):6.2f}")
    for x in range(8):
        e = end[np.arange(len(st))[ran] % 8 == x]
        if len(e):
            print(f"  XCD slot {x}: {len(e)} workgroups, end min {e.min():.1f} mean {e.mean():.1f} max {e.max():.1f}")


if __name__ == "__main__":
    main()
