"""CPU: `relabel_image_stardist` / `relabel_image_stardist3D` (stardist/geometry/geom2d.py:200-211, geom3d.py:201-217) and the package's
top-level names.  Goldens: tests/golden/relabel_reference.npz, made by the reference's OWN two functions with the real scikit-image
regionprops / polygon and the compiled reference natives (tests/golden/make_relabel_golden.py).  Here the product's host logic runs with
the oracle standing in for the HIP natives (the GPU suite runs the same cases on the device: tests/test_gpu_relabel.py)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "relabel_reference.npz"))


@pytest.mark.parametrize("dim,k", [(2, k) for k in range(int(G["n2d"]))] + [(3, k) for k in range(int(G["n3d"]))])
def test_region_centroids_equal_real_regionprops(dim, k):
    """labels in ascending order and the float64 centroids bit for bit (so that the truncation to the centre pixel cannot differ)"""
    from stardist_amd.geometry.geom2d import _region_centroids
    labs, cen = _region_centroids(G["in%dd_%d" % (dim, k)])
    assert np.array_equal(labs, G["lab%dd_%d" % (dim, k)])
    assert cen.dtype == np.float64 and np.array_equal(cen, G["cen%dd_%d" % (dim, k)])


def _oracle():
    from oracle import port, ref
    if not ref.available():
        pytest.skip("oracle/_ref not built")
    ref.set_threads(1)
    return port


@pytest.mark.parametrize("k", range(int(G["n2d"])))
def test_relabel_image_stardist_host_logic_equals_reference(k, monkeypatch):
    port = _oracle()
    from stardist_amd.geometry import geom2d
    monkeypatch.setattr(geom2d, "star_dist", lambda lbl, n_rays, **kw: port.star_dist(lbl, n_rays))
    monkeypatch.setattr(geom2d, "polygons_to_label", lambda dist, points, shape: port.polygons_to_label(dist, points, shape))
    lbl = G["in2d_%d" % k]
    out = geom2d.relabel_image_stardist(lbl, int(G["rays2d_%d" % k]))
    assert out.shape == lbl.shape and np.array_equal(np.asarray(out).astype(np.int32), G["out2d_%d" % k]), str(G["name2d_%d" % k])


@pytest.mark.parametrize("k", range(int(G["n3d"])))
def test_relabel_image_stardist3d_host_logic_equals_reference(k, monkeypatch):
    port = _oracle()
    from stardist_amd.geometry import geom3d
    from stardist_amd.rays3d import Rays_GoldenSpiral
    monkeypatch.setattr(geom3d, "star_dist3D", lambda lbl, rays, **kw: port.star_dist3D(lbl, rays.vertices))
    monkeypatch.setattr(geom3d, "polyhedron_to_label",
                        lambda dist, points, rays, shape, labels=None, verbose=False: port.polyhedron_to_label(
                            dist, points, rays.vertices, rays.faces, shape, labels=labels, verbose=verbose))
    lbl = G["in3d_%d" % k]
    rays = Rays_GoldenSpiral(int(G["rays3d_%d" % k]), anisotropy=tuple(1.0 / G["eps3d_%d" % k]))
    out = geom3d.relabel_image_stardist3D(lbl, rays)
    assert np.array_equal(np.asarray(out).astype(np.int32), G["out3d_%d" % k]), str(G["name3d_%d" % k])
    assert set(np.unique(out)) - {0} <= set(int(v) for v in G["lab3d_%d" % k])         # the regions' own ids (geom3d.py:216)


def test_relabel_refuses_what_the_reference_refuses():
    from stardist_amd.geometry import relabel_image_stardist, relabel_image_stardist3D
    from stardist_amd.rays3d import Rays_GoldenSpiral
    with pytest.raises(ValueError):
        relabel_image_stardist(np.zeros((4, 4, 4), np.uint16), 32)              # "lbl image should be 2 dimensional"
    with pytest.raises(ValueError):
        relabel_image_stardist3D(np.zeros((4, 4), np.uint16), Rays_GoldenSpiral(8))
    with pytest.raises(ValueError):
        relabel_image_stardist(np.zeros((4, 4), np.float32), 32)                # not an integer label array
    with pytest.raises(ValueError):
        relabel_image_stardist(-np.ones((4, 4), np.int32), 32)


def test_top_level_names_follow_the_reference_package():
    """what `from stardist import ...` offers on the prediction path (stardist/__init__.py:12-20) resolves under the same names"""
    import stardist_amd as s
    names = ["non_maximum_suppression", "non_maximum_suppression_3d", "non_maximum_suppression_3d_sparse", "edt_prob", "export_imagej_rois",
             "fill_label_holes", "sample_points", "calculate_extents", "gputools_available",
             "star_dist", "polygons_to_label", "relabel_image_stardist", "ray_angles", "dist_to_coord", "star_dist3D", "polyhedron_to_label",
             "relabel_image_stardist3D", "rays_from_json", "Rays_Cartesian", "Rays_SubDivide", "Rays_Tetra", "Rays_Octo", "Rays_GoldenSpiral",
             "Rays_Explicit"]
    for n in names:
        assert callable(getattr(s, n)), n
        assert n in dir(s)
    with pytest.raises(AttributeError):
        s.no_such_name
