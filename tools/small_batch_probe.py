#!/usr/bin/env python3
"""lbm_step(n) for small n on the 256^3 wind tunnel: MLUPS of calls of n steps back to back, no fields read in between
(an even n runs in fused pairs to the end; an odd one ends on a single step).

    python tools/small_batch_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    from cfd_amd import lbm
    W = H = D = 256
    s = lbm.Solver(W, H, D, preset=lbm.PRESET_WIND_TUNNEL)
    s.init()
    s.step(200)                      # the library settles on the form of its first-step loads
    for n in (1, 2, 3, 4, 5, 6, 10, 11, 20, 100):
        calls = max(4, 600 // n)
        for _ in range(3):
            s.step(n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            s.step(n)
        s.sync()
        dt = time.perf_counter() - t0
        print(f"n = {n:3d}: {dt / (calls * n) * 1e3:.4f} ms per step, {W * H * D * calls * n / dt / 1e6:8.0f} MLUPS", flush=True)
    s.close()


if __name__ == "__main__":
    main()
