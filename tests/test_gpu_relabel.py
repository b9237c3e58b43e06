"""GPU: `relabel_image_stardist` / `relabel_image_stardist3D` (stardist/geometry/geom2d.py:200-211, geom3d.py:201-217) on the HIP natives
against label images produced by the reference's OWN two functions (real scikit-image regionprops / polygon, compiled reference
star_dist / polyhedron rasteriser: tests/golden/make_relabel_golden.py), and the reference's consistency tests for them
(tests/test_stardist2D.py:46-56, tests/test_stardist3D.py:55-66).  The host logic is also pinned on the CPU with the oracle standing in
for the natives (tests/test_cpu_relabel.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "relabel_reference.npz"))


@pytest.mark.parametrize("k", range(int(G["n2d"])))
def test_relabel_image_stardist_equals_reference(k):
    from stardist_amd import relabel_image_stardist
    lbl = G["in2d_%d" % k]
    out = relabel_image_stardist(lbl, int(G["rays2d_%d" % k]))
    assert out.shape == lbl.shape
    assert np.array_equal(np.asarray(out).astype(np.int32), G["out2d_%d" % k]), str(G["name2d_%d" % k])


def _on_hull_boundary(voxels, lbl, rays):
    """does the voxel lie ON the convex hull of one of the star polyhedra relabel_image_stardist3D paints?  (tests/_hull.py: the one
    documented deviation of the 3D rasteriser)"""
    from _hull import on_hull_boundary
    from stardist_amd import star_dist3D
    from stardist_amd.geometry.geom2d import _region_centroids
    labs, cen = _region_centroids(lbl)
    pts = cen.astype(int)
    dist = np.maximum(np.asarray(star_dist3D(lbl, rays))[tuple(pts.T)].reshape(len(pts), len(rays)), 1e-3).astype(np.float32)
    return on_hull_boundary(voxels, pts, dist, rays.vertices)


@pytest.mark.parametrize("k", range(int(G["n3d"])))
def test_relabel_image_stardist3d_equals_reference(k):
    """voxel for voxel, except voxels that lie exactly on the hull of their polyhedron (lattice-aligned synthetic shapes put the tips of the
    axis-aligned rays exactly on voxels): there the reference's own answer is rounding noise of Qhull's planes -- at most two per object"""
    from stardist_amd import Rays_GoldenSpiral, relabel_image_stardist3D
    lbl = G["in3d_%d" % k]
    rays = Rays_GoldenSpiral(int(G["rays3d_%d" % k]), anisotropy=tuple(1.0 / G["eps3d_%d" % k]))
    out = np.asarray(relabel_image_stardist3D(lbl, rays)).astype(np.int32)
    assert out.shape == lbl.shape
    diff = np.argwhere(out != G["out3d_%d" % k])
    if len(diff):
        assert len(diff) <= 2 * len(G["lab3d_%d" % k]), (str(G["name3d_%d" % k]), len(diff))
        assert _on_hull_boundary(diff, lbl, rays).all(), (str(G["name3d_%d" % k]), diff[:8])


def _circle_image(shape, radius, eps):
    xs = tuple(np.arange(s) - s // 2 for s in shape)
    Xs = np.meshgrid(*xs, indexing="ij")
    return (np.sqrt(np.sum([X ** 2 / e ** 2 for X, e in zip(Xs, eps)], axis=0)) < radius).astype(np.uint16)


@pytest.mark.parametrize("n_rays", (32, 64))
@pytest.mark.parametrize("eps", ((1, 1), (.4, 1.3)))
def test_relabel_consistency_2d(n_rays, eps):
    """tests/test_stardist2D.py:46-56: an already star-convex label image gets (almost) perfectly relabelled"""
    from stardist_amd import relabel_image_stardist
    lbl1 = _circle_image((32, 32), 8, eps)
    lbl2 = relabel_image_stardist(lbl1, n_rays)
    assert 1 - np.count_nonzero((lbl1 > 0) & (lbl2 > 0)) / np.count_nonzero(lbl1 > 0) < 1e-1


@pytest.mark.parametrize("n_rays", (64, 128))
@pytest.mark.parametrize("eps", ((1, 1, 1), (.4, 1.3, .7)))
def test_relabel_consistency_3d(n_rays, eps):
    """tests/test_stardist3D.py:55-66"""
    from stardist_amd import Rays_GoldenSpiral, relabel_image_stardist3D
    rays = Rays_GoldenSpiral(n_rays, anisotropy=1. / np.array(eps))
    lbl1 = relabel_image_stardist3D(_circle_image((32, 32, 32), 8, eps), rays)
    lbl2 = relabel_image_stardist3D(lbl1, rays)
    assert 1 - np.count_nonzero((lbl1 > 0) & (lbl2 > 0)) / np.count_nonzero(lbl1 > 0) < 1e-1
