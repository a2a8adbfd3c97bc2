"""The C++ `lbm_solver` mirror (cfd_amd/csrc/lbm_solver.hpp) and the headless driver, built by
`make -C cfd_amd/csrc` (g++ against the C ABI), run on the GPU."""
import os
import subprocess

import pytest

from conftest import CUBE_OBJ, ROOT

pytestmark = pytest.mark.gpu
CSRC = os.path.join(ROOT, "cfd_amd", "csrc")


def test_cpp_lbm_solver_tests():
    exe = os.path.join(CSRC, "test_lbm_solver")
    assert os.path.exists(exe), "build it with make -C cfd_amd/csrc"
    r = subprocess.run([exe, CUBE_OBJ], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all C++ lbm_solver tests passed" in r.stdout


def test_headless_driver_known_answer(tmp_path):
    exe = os.path.join(CSRC, "lbm_headless")
    prefix = str(tmp_path / "out")
    r = subprocess.run([exe, "--dims", "64", "64", "64", "--steps", "10", "--obj", CUBE_OBJ, "--dump", prefix],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "solid=9196" in r.stdout and "sum_rho=255103.576208234" in r.stdout, r.stdout
    assert os.path.getsize(prefix + ".u.raw") == 64 ** 3 * 16
    assert os.path.getsize(prefix + ".solid.raw") == 64 ** 3


@pytest.mark.parametrize("extra", [[], ["--scheme", "aa"], ["--scheme", "slide"], ["--preset", "cavity"], ["--autotune", "--steps", "120"],
                                   ["--single-steps"], ["--copy-engine"], ["--copy-engine", "--scheme", "aa"], ["--copy-engine", "--preset", "cavity"]])
def test_native_rccl_slab_driver(extra):
    """tools/lbm_slab_rccl.cpp: C++ host, RCCL ncclSend/ncclRecv between the slabs of one process.
    On this one-GPU box the wind-tunnel ring sends to itself through RCCL; --check compares with the
    undecomposed run bit for bit.  --copy-engine: the same exchange as peer copies ordered by events (no RCCL kernels)."""
    exe = os.path.join(CSRC, "lbm_slab_rccl")
    assert os.path.exists(exe), "build it with make -C cfd_amd/csrc"
    r = subprocess.run([exe, "--dims", "96", "48", "40", "--gpus", "1", "--steps", "25", "--obj", CUBE_OBJ, "--check"] + extra,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bit-identical" in r.stdout, r.stdout
    # A/B slabs run all but the last step of a batch as pairs of steps (lbm_pair_*), everything else single steps
    pairs = int(r.stdout.split(" pairs of steps")[0].rsplit("(", 1)[1])
    if "--single-steps" in extra or "--scheme" in extra:
        assert pairs == 0
    elif not extra:
        assert pairs == 7, r.stdout          # 25 steps = 10 warm-up + 15 timed = 7 pairs + 1 single step
