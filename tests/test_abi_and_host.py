"""CPU-side checks of the product library: it loads, exports every symbol include/lbm_abi.h
declares, refuses to compute without a device, and its host-only stages (voxeliser, per-cell
arithmetic) agree with the oracle bit for bit."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import CUBE_OBJ, GOLDEN, ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "lbm_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lbm_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lbm):
    L = lbm.load_library()
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/lbm_abi.h but not exported"
    assert sorted(lbm.SIGNATURES) == names
    assert L.lbm_abi_version() == lbm.ABI_VERSION == 3


def test_no_silent_cpu_fallback(lbm):
    """Without a HIP device the product path must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the failure path is exercised on the CPU-only runner")
    with pytest.raises(lbm.LbmError) as e:
        lbm.Solver(8, 8, 8)
    assert e.value.code == -2 and "no CPU path" in str(e.value)


def test_product_does_not_touch_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "cfd_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                text = open(os.path.join(root, f)).read()
                assert "lbm_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_folded_collision_is_bit_exact(oracle, lbm):
    """cfd_amd/csrc/d3q19.h against the literal loops of lbm_solver.cu:320-401."""
    rng = np.random.default_rng(12345)
    w = oracle.tables()["w"]
    for i in range(4000):
        eps = rng.uniform(-0.05, 0.05, 19)
        f = (w * (1 + eps)).astype(np.float32)
        if i % 500 == 0:
            f[:] = 0          # rho <= 1e-9 "safety" branch (Q5)
        ws, wb = (1.98, 1.90) if i % 3 else (float(rng.uniform(0.6, 1.99)), float(rng.uniform(0.6, 1.99)))
        a, ra = oracle.collide_cell(f, ws, wb)
        b, rb = lbm.host_collide_cell(f, ws, wb)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (i, f)
        assert np.array_equal(ra, rb)


@pytest.mark.parametrize("obj", ["cube.obj", "car_like.obj"])
@pytest.mark.parametrize("dims", [(64, 64, 64), (128, 64, 64), (96, 40, 48), (33, 17, 70)])
def test_product_voxeliser_matches_oracle(oracle, lbm, obj, dims):
    path = os.path.join(GOLDEN, obj)
    mo, so, rco = oracle.voxelize_obj(path, *dims)
    mp, sp, rcp = lbm.host_voxelize_obj(path, *dims)
    assert rco == 0 and rcp == 0 and np.float32(so) == np.float32(sp)
    np.testing.assert_array_equal(mp, mo)
    assert set(np.unique(mp)) <= {0, 255}


def test_product_voxeliser_known_answer(lbm):
    m, scale, rc = lbm.host_voxelize_obj(CUBE_OBJ, 64, 64, 64)
    assert rc == 0 and int((m > 0).sum()) == 9196


def test_voxeliser_random_soup_and_obj_syntax(oracle, lbm, tmp_path):
    """Open triangle soup (odd hit counts, grazing hits), mixed index syntax, degenerate faces."""
    rng = np.random.default_rng(5)
    p = tmp_path / "soup.obj"
    with open(p, "w") as fh:
        fh.write("# soup\n\n")
        for _ in range(60):
            x, y, z = rng.uniform(-5, 5, 3)
            fh.write(f"v {x:.5f} {y:.5f} {z:.5f}\n")
        fh.write("vt 0 0\nvn 0 0 1\n")
        for i in range(40):
            a, b, c, d = rng.integers(1, 61, 4)
            if i % 4 == 0:
                fh.write(f"f {a} {b} {c}\n")
            elif i % 4 == 1:
                fh.write(f"f {a}/1/1 {b}/1/1 {c}/1/1 {d}/1/1\n")
            elif i % 4 == 2:
                fh.write(f"f {a - 61} {b - 61} {c - 61}\n")
            else:
                fh.write(f"f {a}//1 {b}//1 {c}//1 {d}//1\n")
        fh.write("f 1 2\nf 1 2 999\n")                  # degenerate / out-of-range faces are skipped
    for dims in [(48, 48, 48), (80, 20, 31)]:
        mo, so, _ = oracle.voxelize_obj(str(p), *dims)
        mp, sp, _ = lbm.host_voxelize_obj(str(p), *dims)
        np.testing.assert_array_equal(mp, mo)
    mo, _, rco = oracle.voxelize_obj(str(tmp_path / "missing.obj"), 16, 8, 8)
    mp, _, rcp = lbm.host_voxelize_obj(str(tmp_path / "missing.obj"), 16, 8, 8)
    assert rco != 0 and rcp == -4
    np.testing.assert_array_equal(mp, mo)               # floor only (lbm_solver.cu:522-524, 549-553)
    assert (mp.reshape(8, 8, 16)[:, 0, :] == 255).all() and int((mp > 0).sum()) == 16 * 8


L_PRISM = os.path.join(GOLDEN, "l_prism.obj")


@pytest.mark.parametrize("dims", [(64, 64, 64), (96, 40, 48), (50, 70, 33)])
def test_concave_ngon_faces_are_ear_clipped(oracle, lbm, dims):
    """Faces with more than four vertices: the reference's loader cuts ears (tiny_obj_loader.h:1751-1973).
    tests/golden/l_prism.obj has concave 7-gon caps whose first corner does not see the whole polygon (a fan
    from it would cover the notch) and whose first three vertices are collinear.  Product == oracle bit for
    bit, and the voxelised solid is the L: filled arms, empty notch."""
    W, H, D = dims
    mo, so, rco = oracle.voxelize_obj(L_PRISM, W, H, D)
    mp, sp, rcp = lbm.host_voxelize_obj(L_PRISM, W, H, D)
    assert rco == 0 and rcp == 0 and np.float32(so) == np.float32(sp)
    np.testing.assert_array_equal(mp, mo)
    # both loaders turn the 7-gon caps into 5 triangles each: 10 + 6 quads x 2
    tris = oracle.mesh_triangles(L_PRISM)
    assert len(tris) == 22
    # geometry after load_mesh: rotate_y(-90) maps (x, y, z) -> (-z, y, x); the mesh is centred in x and z and
    # sits on y = 0.  The L occupies z' = x in [0, 40] for y < 10 and z' in [0, 10] for y in [10, 30).
    m = mp.reshape(D, H, W) > 0
    s = np.float32(sp)
    zc = D / 2.0
    z_of = lambda xo: int(np.floor(zc + (xo - 20.0) * s))          # original x -> grid z
    x_mid = W // 2
    y_low, y_high = max(1, int(5 * s)), int(20 * s)
    assert m[z_of(30.0), y_low, x_mid] and m[z_of(5.0), y_low, x_mid]          # the bar, both ends
    assert m[z_of(5.0), y_high, x_mid]                                          # the tall arm
    assert not m[z_of(30.0), y_high, x_mid]                                     # the notch stays empty
    # solid volume ~ the L's area x depth (in voxels), far from the 1.5 x a fan over the notch would give
    vol = m[:, 1:, :].sum()
    expect = (40 * 10 + 10 * 20) * 30 * float(s) ** 3
    assert 0.75 * expect < vol < 1.2 * expect, (vol, expect)     # voxel rounding: up to a voxel per face on the coarse grids
