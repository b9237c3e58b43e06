"""CPU: the oracle (compiled reference in oracle/_ref + numpy port) against the committed golden vectors
(tests/golden/*.npz, produced by tests/golden/make_golden.py from the reference itself)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "reference_outputs.npz"))


@pytest.mark.parametrize("H,W,R,thr", [(256, 256, 32, 0.4), (512, 512, 32, 0.4), (356, 299, 11, 0.5), (114, 217, 32, 0.3)])
def test_ref_nms2d_matches_golden(refmods, H, W, R, thr):
    from oracle import synth
    d, p, s = synth.s2d_uniform(H, W, n_rays=R)
    keep = refmods.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr))
    n, m = G["nms2d_%d_%d_%d_n" % (H, W, R)]
    assert (len(d), int(keep.sum())) == (n, m)
    assert np.array_equal(np.packbits(keep), G["nms2d_%d_%d_%d_keep" % (H, W, R)])


def test_survey_calibration_counts():
    """S2D-uniform 512^2 -> 26 010 candidates, 1 622 survivors"""
    assert tuple(G["nms2d_512_512_32_n"]) == (26010, 1622)


def test_ref_old_equals_new_nms(refmods):
    """the reference's own cross-check tests/test_nms2D.py:78-110 (old grid-search NMS == new kd-tree NMS),
    restated at the native boundary: same survivors for the same seeded candidates"""
    from oracle import port, synth
    m = refmods.stardist2d()
    for shape, R in (((356, 299), 11), ((114, 217), 32)):
        dist, prob = synth.s2d_uniform(shape[0], shape[1], n_rays=R, dense=True)
        mask = port.ind_prob_thresh(prob, 0.9, b=2)
        pts = np.stack(np.where(mask), 1)
        d, s = dist[mask], prob[mask]
        ind = np.argsort(s)[::-1]
        d, s, pts = d[ind], s[ind], pts[ind]
        keep_new = m.c_non_max_suppression_inds(np.ascontiguousarray(d), np.ascontiguousarray(pts.astype(np.float32)), 1, 1, 0, np.float32(0.4))
        coord = port.dist_to_coord(d, pts)                       # (n,2,R)
        mapping = -np.ones(mask.shape, np.int32)
        mapping.flat[np.flatnonzero(mask)[ind]] = range(len(ind))
        keep_old = m.c_non_max_suppression_inds_old(np.ascontiguousarray(coord.astype(np.int32)), mapping, np.float32(0.4),
                                                    np.int32(1), np.int32(1), np.int32(1), np.int32(0))
        assert keep_new.sum() == keep_old.sum()
        assert np.array_equal(keep_new, keep_old)


def test_ref_star_dist_matches_golden(refmods):
    from oracle import synth
    lbl, _, _ = synth.s2d_nuclei_labels(200, 231, seed=3)
    m = refmods.stardist2d()
    assert np.array_equal(m.c_star_dist(lbl, 32, 1, 1)[::7, ::7], G["stardist2d_32_1"])
    assert np.array_equal(m.c_star_dist(lbl, 17, 2, 2)[::7, ::7], G["stardist2d_17_2"])
    # the reference's own properties: dtype invariance (tests/test_stardist2D.py:7-17), grid == subsample (:68-80)
    a = m.c_star_dist(lbl.astype(np.uint16), 32, 1, 1)
    assert np.array_equal(a[::2, ::2], m.c_star_dist(lbl, 32, 2, 2))


def test_ref_3d_matches_golden(refmods):
    from oracle import synth
    from stardist_amd.rays3d import Rays_GoldenSpiral
    rays = Rays_GoldenSpiral(96)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    d, p, s, nobj = synth.s3d_nuclei(64, V)
    m = refmods.stardist3d()
    keep = m.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(0.3))
    assert (len(d), int(keep.sum()), nobj) == tuple(G["nms3d_64_n"])
    assert np.array_equal(np.packbits(keep), G["nms3d_64_keep"])
    lbl = m.c_polyhedron_to_label(d[keep], p[keep], V, F, np.arange(1, keep.sum() + 1, dtype=np.int32), 0, 0, 0, 0, (64, 64, 64))
    assert np.array_equal(np.bincount(lbl.ravel()), G["raster3d_64_hist"])


def test_clipper_probe_matches_golden(refmods):
    a = np.array([refmods.clipper_area(G["clip_xa"][i], G["clip_ya"][i], G["clip_xb"][i], G["clip_yb"][i]) for i in range(64)], np.float32)
    assert np.array_equal(a, G["clip_area"])


def test_rays_match_reference_golden():
    """stardist_amd.rays3d against vertices/faces produced by the reference's own rays3d.py"""
    from stardist_amd import rays3d as M
    R = np.load(os.path.join(ROOT, "tests", "golden", "rays_reference.npz"))
    for name, obj in [("gs96", M.Rays_GoldenSpiral(96)), ("gs32", M.Rays_GoldenSpiral(32)), ("gs65", M.Rays_GoldenSpiral(65)),
                      ("gs96a", M.Rays_GoldenSpiral(96, anisotropy=(2, 1, 1))), ("cart", M.Rays_Cartesian(11, 5)),
                      ("tetra3", M.Rays_Tetra(3)), ("octo2", M.Rays_Octo(2))]:
        assert np.array_equal(obj.vertices, R[name + "_v"]), name
        assert np.array_equal(obj.faces, R[name + "_f"]), name
    assert M.rays_from_json(M.Rays_GoldenSpiral(96, anisotropy=(2, 1, 1)).to_json()).vertices.shape == (96, 3)


def test_port_polygon_rule():
    """the restated scikit-image rule: label i == polygon mask (cf. tests/test_big.py:202-213), vertices and
    edge pixels count as inside, painting order = later overwrites earlier"""
    from oracle import port
    r = np.array([2.0, 2.0, 8.0, 8.0]); c = np.array([3.0, 9.0, 9.0, 3.0])
    rr, cc = port.polygon(r, c, (12, 12))
    m = np.zeros((12, 12), bool); m[rr, cc] = True
    assert m[2:9, 3:10].all() and m.sum() == 49          # closed square incl. its edges
    coord = np.stack([np.stack([r, c]), np.stack([r + 1, c + 1])])
    lbl = port.polygons_to_label_coord(coord, (12, 12))
    assert lbl[2, 3] == 1 and lbl[5, 5] == 2 and lbl[9, 10] == 2


def test_dist_to_volume_and_centroid_tensor_formulation_vs_reference(refmods):
    """the tensor formulation used by stardist_amd.lib.stardist3d.c_dist_to_volume / c_dist_to_centroid (run here on CPU
    tensors; the public wrappers only run on a HIP device) against the compiled reference, within the 1e-5 float tolerance"""
    import torch
    from stardist_amd.lib import stardist3d as sd3
    from stardist_amd.rays3d import Rays_GoldenSpiral
    rays = Rays_GoldenSpiral(33, anisotropy=(2, 1, 1))
    V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
    rng = np.random.RandomState(0)
    dist = (5 + 3 * rng.rand(5, 6, 7, 33)).astype(np.float32)
    dist[0, 0, 0] = 0
    m3 = refmods.stardist3d()
    ref_vol = m3.c_dist_to_volume(dist, V, F)
    tv, tf = torch.from_numpy(V), torch.from_numpy(F.astype(np.int64))
    vol = sd3._dist_to_volume_t(torch.from_numpy(dist), tv, tf, chunk=2).numpy()
    assert vol.shape == ref_vol.shape and np.allclose(vol, ref_vol, rtol=1e-5, atol=1e-4)
    for absolute in (0, 1):
        ref_c = m3.c_dist_to_centroid(dist, V, F, absolute)
        c = sd3._dist_to_centroid_t(torch.from_numpy(dist), tv, tf, absolute, chunk=3).numpy()
        assert c.shape == ref_c.shape and np.allclose(c, ref_c, rtol=1e-5, atol=1e-4)


RASTER_CASES = ("stars32", "stars8_small", "stars64_big", "stars5_int")


def _raster_golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raster2d_reference.npz"))


@pytest.mark.parametrize("name", RASTER_CASES)
def test_port_rasteriser_equals_reference_with_real_skimage(name):
    """oracle/port.py's restatement of skimage.draw.polygon + geom2d.polygons_to_label against golden label images made by the
    reference's own code running on the real scikit-image (tests/golden/make_raster2d_golden.py): bit-identical"""
    from oracle import port
    g = _raster_golden()
    lbl = port.polygons_to_label(g[name + "_dist"], g[name + "_points"], tuple(g[name + "_shape"]), prob=g[name + "_prob"], thr=0.2)
    assert lbl.dtype == np.int32 and np.array_equal(lbl, g[name + "_labels"])


def test_port_polygon_rule_on_lattice_and_degenerate_cases():
    """vertices / edges exactly on pixel centres, half-integer rectangles, zero-area, bow-tie, clipped polygons"""
    from oracle import port
    g = _raster_golden()
    coord, shape = g["explicit_coord"], tuple(g["explicit_shape"])
    assert np.array_equal(port.polygons_to_label_coord(coord, shape), g["explicit_labels"])
    for i, c in enumerate(coord):
        rr, cc = port.polygon(c[0], c[1], shape)
        m = np.zeros(shape, bool); m[rr, cc] = True
        assert np.array_equal(m, g["explicit_mask%d" % i]), i


def test_glue_functions_equal_reference_golden():
    """relabel_sequential (matching.py:319-408) and _ind_prob_thresh (nms.py:6-17): product mirror and oracle port against outputs
    of the reference's own functions (tests/golden/make_glue_golden.py)"""
    from oracle import port
    from stardist_amd.matching import relabel_sequential
    from stardist_amd.nms import _ind_prob_thresh
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glue_reference.npz"))
    for k in "abcde":
        lab, off = g["relabel_%s_in" % k], int(g["relabel_%s_offset" % k])
        for fn in (relabel_sequential, port.relabel_sequential):
            r, fw, inv = fn(lab.copy(), off)
            assert r.dtype == g["relabel_%s_out" % k].dtype and np.array_equal(r, g["relabel_%s_out" % k]), (k, fn.__module__)
            assert np.array_equal(np.asarray(fw), g["relabel_%s_fw" % k]) and np.array_equal(np.asarray(inv), g["relabel_%s_inv" % k])
    for k in ("2d", "3d", "b0", "bt"):
        prob, thr, b = g["thresh_%s_prob" % k], float(g["thresh_%s_args" % k][0]), g["thresh_%s_b" % k]
        b = None if (b.ndim == 0 and int(b) == -1) else (int(b) if b.ndim == 0 else tuple(map(tuple, b)))
        assert np.array_equal(np.asarray(_ind_prob_thresh(prob, thr, b=b)), g["thresh_%s_mask" % k])
        assert np.array_equal(port.ind_prob_thresh(prob, thr, b=b), g["thresh_%s_mask" % k])


def test_oracle_edt_prob_equals_reference_function():
    """oracle.port.edt_prob (exhaustive nearest-other-label search) against goldens made by the reference's own _edt_prob_scipy
    (tests/golden/make_utils_golden.py): the checker the GPU kernel is compared with on larger inputs"""
    import warnings
    from oracle import port
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "utils_reference.npz"))
    assert np.array_equal(port.edt_prob(g["edt_lab2"]), g["edt_prob2"])
    assert np.array_equal(port.edt_prob(g["edt_lab3"], anisotropy=(2.0, 1.0, 1.0)), g["edt_prob3"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert np.array_equal(port.edt_prob(g["edt_const"]), g["edt_prob_const"])


def test_imagej_roi_export_equals_reference_functions(tmp_path):
    """polyroi_bytearray / export_imagej_rois: goldens made by the reference's own functions (tests/golden/make_utils_golden.py)"""
    import zipfile
    from stardist_amd import utils
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "utils_reference.npz"))
    polys = g["roi_polys"]
    assert bytes(utils.polyroi_bytearray(polys[0][1], polys[0][0], pos=3, subpixel=True)) == g["roi_bytes_sub"].tobytes()
    assert bytes(utils.polyroi_bytearray(polys[1][1], polys[1][0], pos=None, subpixel=False)) == g["roi_bytes_int"].tobytes()
    utils.export_imagej_rois(str(tmp_path / "rois.zip"), [polys[:2], polys[2:]])
    with zipfile.ZipFile(str(tmp_path / "rois.zip")) as z:
        names = sorted(z.namelist())
        assert names == list(g["roi_zip_names"])
        assert b"".join(z.read(n) for n in names) == g["roi_zip_concat"].tobytes()


def test_reference_nms2d_is_not_translation_invariant(refmods):
    """A property of the REFERENCE that bounds what 'big == whole' can mean in 2D: c_non_max_suppression_inds computes the polygon
    vertices as float32 `p + d * cos/sin` at absolute image coordinates (stardist2d.cpp:453-455) and truncates them to the integer
    lattice (:471), so the same candidates shifted by a few thousand pixels give slightly different integer polygons and a few
    different decisions.  A block of predict_instances_big works in block-local coordinates; the whole image does not."""
    from oracle import synth
    d, p, s = synth.s2d_uniform(1024, 1024)
    m = refmods.stardist2d(); refmods.set_threads(min(os.cpu_count() or 1, 8))
    k0 = m.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4))
    k1 = m.c_non_max_suppression_inds(d, (p + np.float32(2048)).astype(np.float32), 1, 1, 0, np.float32(0.4))
    ndiff = int((k0 != k1).sum())
    assert 0 < ndiff < 1e-3 * len(d), ndiff                        # measured: 40 of 104 580


def test_reference_cascade_on_rays_cartesian_runs_on_its_error_paths(refmods, capfd):
    """What the reference does with Rays_Cartesian (pole rays 1e-12 apart: one float32 point, or points on one line through the centre), stage by
    stage with its own functions (oracle shim `pair_cascade`, stardist3d_impl.cpp:1207-1318) -- the behaviour the device NMS follows since
    round 6 (tests/test_gpu_lattice.py): the inner-sphere bound is ~0 (a degenerate face has distance 0), the kernel stage returns Qhull's
    error value for EVERY pair (zero-normal half-spaces), the hull stage returns 1e10 whenever the midpoint of the centres is not inside
    both hulls, and the rendered overlap counts lattice points far outside a polyhedron (a zero-volume tetrahedron passes the four
    `det >= 0` tests on its whole plane): more voxels "inside" than the polyhedron has volume."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_lattice_golden import rays_of
    from oracle import synth
    rays = rays_of("cartesian_8_5")
    V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
    d, p, s = synth.lattice_candidates_3d(len(V), "int", size=48)
    P = p.astype(np.float32)
    pairs = np.array([[3, 8], [3, 15], [22, 27], [22, 29], [2, 21]], np.int32)
    C = refmods.pair_cascade(d, P, V, F, pairs)
    capfd.readouterr()                                   # (Qhull's precision warnings)
    rk, rh = refmods.pair_volumes(d, P, V, F, pairs)
    capfd.readouterr()
    assert np.array_equal(C[:, 4], rk) and np.array_equal(C[:, 5], rh)
    assert (C[:, 4] == 0).all()                          # kernel stage: error value for every pair
    assert (np.abs(C[:, 3]) < 1e-6).all()                # inner spheres: nothing
    assert (C[:, 5] > 1e9).sum() >= 3 and (C[:, 5] < 1e9).sum() >= 1
    assert (C[:, 2] > 0).all() and (C[:, 0] > 200).all() and (C[:, 1] > 200).all()
    # the rendered overlap of (3, 8) is what suppresses candidate 8 at threshold 0.2 in the lattice golden: 65 voxels of 310.2
    assert int(C[0, 6]) == 65 and abs(C[0, 0] - 310.2327) < 1e-3
    G3 = np.load(os.path.join(ROOT, "tests", "golden", "lattice_reference.npz"))
    keep = np.unpackbits(G3["nms3d_cartesian_8_5_int_0.2"])[:len(d)].astype(bool)
    assert keep[3] and not keep[8]


@pytest.mark.parametrize("name", ["octo1", "tetra3"])
def test_ref_3d_matches_lattice_golden_more(refmods, name, capfd):
    """the round-6 lattice goldens of the symmetric ray sets (tests/golden/make_lattice_golden_more.py) are what the compiled reference
    returns here: keep flags at both thresholds, one OpenMP thread"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_lattice_golden_more import rays_of_more
    from oracle import synth
    G4 = np.load(os.path.join(ROOT, "tests", "golden", "lattice_reference_more.npz"))
    rays = rays_of_more(name)
    V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
    refmods.set_threads(1)
    for fam in ("const", "int"):
        d, p, s = synth.lattice_candidates_3d(len(V), fam, size=48)
        for thr in (0.2, 0.4):
            keep = refmods.stardist3d().c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(thr)).astype(bool)
            assert np.array_equal(np.packbits(keep), G4["nms3d_%s_%s_%.1f" % (name, fam, thr)]), (name, fam, thr)
    capfd.readouterr()
