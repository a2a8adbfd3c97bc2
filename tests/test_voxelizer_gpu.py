"""Device voxeliser (lbm_load_mesh: ray casting on the GPU, straight into the bit-packed mask)
against the host voxeliser of the same library and against the oracle's restatement of
mesh_voxelizer.cpp:146-216 / lbm_solver.cu:521-556.  Bar: the mask is bit-exact."""
import os

import numpy as np
import pytest

from conftest import CUBE_OBJ, GOLDEN

pytestmark = pytest.mark.gpu
CAR_OBJ = os.path.join(GOLDEN, "car_like.obj")


@pytest.mark.parametrize("obj", [CUBE_OBJ, CAR_OBJ, os.path.join(GOLDEN, "l_prism.obj")])   # l_prism: concave 7-gon faces (ear clipping)
@pytest.mark.parametrize("dims", [(64, 64, 64), (400, 100, 100), (128, 64, 64), (70, 33, 17), (200, 300, 40)])
def test_device_mask_equals_host_and_oracle(oracle, lbm, obj, dims):
    W, H, D = dims
    g = lbm.Solver(W, H, D)
    g.add_solid(1, 1, 1, 3, 3, 3)                      # load_mesh replaces the whole mask
    g.load_mesh(obj)
    dev = g.solid_mask()
    host, scale, rc = lbm.host_voxelize_obj(obj, W, H, D)
    assert rc == 0
    np.testing.assert_array_equal(dev, host)
    o = oracle.Solver(W, H, D)
    o.load_mesh(obj)
    np.testing.assert_array_equal(dev, o.solid_mask())


def test_known_answers_cube(lbm):
    """SURVEY.md Appendix A.1 straight from the device voxeliser."""
    g = lbm.Solver(64, 64, 64)
    g.load_mesh(CUBE_OBJ)
    m = g.solid_mask().reshape(64, 64, 64) > 0
    assert m.sum() == 9196
    body = m.copy(); body[:, 0, :] = False
    zz, yy, xx = np.nonzero(body)
    assert (xx.min(), xx.max(), yy.min(), yy.max(), zz.min(), zz.max()) == (27, 36, 1, 10, 7, 57)


def test_slab_contexts_voxelise_their_own_layers(lbm):
    W, H, Dg = 96, 48, 40
    host, _, _ = lbm.host_voxelize_obj(CAR_OBJ, W, H, Dg)
    host = host.reshape(Dg, H * W)
    z0 = 0
    for d in (13, 20, 7):
        s = lbm.Solver(W, H, d, z0=z0, depth_global=Dg)
        s.load_mesh(CAR_OBJ)
        np.testing.assert_array_equal(s.solid_mask().reshape(d, -1), host[z0:z0 + d])
        s.close()
        z0 += d


def test_many_hits_fall_back_to_host(lbm, tmp_path):
    """A ray through 40 boxes has 80 hits, more than the kernel keeps per ray: the library redoes
    the mask on the host, with the same result."""
    lines, nv = [], 0
    for i in range(40):
        x0, x1 = 10.0 * i, 10.0 * i + 4.0
        vs = [(x, y, z) for x in (x0, x1) for y in (0.0, 5.0) for z in (0.0, 5.0)]
        lines += [f"v {x} {y} {z}" for x, y, z in vs]
        q = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
        lines += ["f " + " ".join(str(nv + a + 1) for a in f) for f in q]
        nv += 8
    path = tmp_path / "boxes.obj"
    path.write_text("\n".join(lines) + "\n")
    W, H, D = 48, 24, 800          # rotate_y(-90) turns the row of boxes along z; rays run along it after scaling
    for dims in ((W, H, D), (800, 24, 48)):
        g = lbm.Solver(*dims)
        g.load_mesh(str(path))
        host, _, _ = lbm.host_voxelize_obj(str(path), *dims)
        np.testing.assert_array_equal(g.solid_mask(), host)
        g.close()
