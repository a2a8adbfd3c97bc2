"""Generates tests/golden/oracle_cases.json: fingerprints of rho / u / solid (and a few sampled
values) after N steps for small cases, computed with the pinned CPU oracle in the authoring
container.  The GPU tests compare the HIP path against these committed fingerprints, so the GPU box
needs nothing from /root/reference.  Usage: python tools/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

CASES = [
    # name, dims, preset, obstacle, steps
    ("cube64_n10", (64, 64, 64), 0, "cube", 10),
    ("cube64_n100", (64, 64, 64), 0, "cube", 100),
    ("empty32_n10", (32, 32, 32), 0, None, 10),
    ("empty128x64x64_n20", (128, 64, 64), 0, None, 20),
    ("car128x64x64_n100", (128, 64, 64), 0, "car", 100),
    ("cavity48_n50", (48, 48, 48), 1, None, 50),
    ("ragged70x33x17_rect_n12", (70, 33, 17), 0, "rect", 12),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(S, name, dims, preset, obstacle, steps, golden_dir):
    s = S(*dims, preset)
    s.init()
    if obstacle == "cube":
        s.load_mesh(os.path.join(golden_dir, "cube.obj"))
    elif obstacle == "car":
        s.load_mesh(os.path.join(golden_dir, "car_like.obj"))
    elif obstacle == "rect":
        W, H, D = dims
        s.add_solid(W // 3, 0, D // 4, W // 5, H // 2, D // 2)
    s.step(steps)
    rho, u, m = s.density(), s.velocity(), s.solid_mask()
    curl = (s.compute_curl(), s.curl_field())[1]
    # normalise -0.0 so fingerprints compare values, not zero signs
    rho = rho + np.float32(0); u = u + np.float32(0); curl = curl + np.float32(0)
    n = rho.size
    samp = np.linspace(0, n - 1, 16).astype(np.int64)
    return {
        "dims": list(dims), "preset": preset, "obstacle": obstacle, "steps": steps,
        "solid_count": int((m > 0).sum()), "solid_sha256": sha((m > 0).astype(np.uint8)),
        "rho_sha256": sha(rho), "u_sha256": sha(u), "curl_sha256": sha(curl),
        "sum_rho": float(rho.astype(np.float64).sum()),
        "umax": float(np.sqrt((u[:, :3].astype(np.float64) ** 2).sum(1)).max()),
        "sample_idx": samp.tolist(), "sample_rho": [float(x) for x in rho[samp]],
        "sample_ux": [float(x) for x in u[samp, 0]],
    }


if __name__ == "__main__":
    g = os.path.join(ROOT, "tests", "golden")
    out = {c[0]: run_case(O.Solver, *c, g) for c in CASES}
    with open(os.path.join(g, "oracle_cases.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in out.items():
        print(k, v["sum_rho"], v["umax"])
