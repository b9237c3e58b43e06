"""GPU: survivor parity (keep ARRAYS, not counts) at the sizes BASELINE.json's metric is quoted on, and the 3D end-to-end
composition (NMS -> polyhedron raster -> relabel incl. negative overlap_label, stardist/models/model3d.py:589-674)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "fullsize_keep.npz"))


def test_nms2d_2048_keep_array_equals_reference(refmods):
    """config 2 (S2D-uniform 2048^2, 416 700 candidates): survivors bit-identical to the compiled reference, run live, and to
    the committed golden bits (tests/golden/make_fullsize_golden.py)"""
    from oracle import synth
    from stardist_amd.lib import stardist2d as sd2
    d, p, s = synth.s2d_uniform(2048, 2048)
    refmods.set_threads(min(os.cpu_count() or 1, 16))          # the 2D reference is thread-count independent
    ref_keep = refmods.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4))
    keep = sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4))
    assert np.array_equal(keep, ref_keep), "mismatching candidates: %s" % np.flatnonzero(keep != ref_keep)[:10]
    g = _golden()
    assert int(g["nms2d_2048_n"]) == len(d)
    assert np.array_equal(np.packbits(keep), g["nms2d_2048_keep"])


def test_nms2d_bench_candidate_set_keep_array_equals_reference(refmods):
    """the bench's own workload: the calibrated U-Net's candidates on the 2048^2 synthetic fluo tile (~4.2e5), NMS on the GPU vs the
    compiled reference on the very same sorted candidate arrays"""
    import torch
    import bench
    from oracle import synth
    from stardist_amd import nms
    from stardist_amd.lib import stardist2d as sd2
    from stardist_amd.models import Config2D, StarDist2D
    dev = torch.device("cuda:0")
    img = torch.from_numpy(synth.s2d_nuclei_image(2048, 2048, seed=0)).to(dev)
    model = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    model.thresholds = dict(prob=0.5, nms=0.4)
    bench.calibrate_heads(model, img)
    prob, dist, points = model.predict_sparse(img)
    assert len(prob) > 300000
    order = nms._argsort_desc(prob)
    d = np.ascontiguousarray(dist[order], np.float32); p = np.ascontiguousarray(points[order], np.float32)
    refmods.set_threads(min(os.cpu_count() or 1, 16))
    ref_keep = refmods.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4))
    keep, stats = sd2.c_non_max_suppression_inds(torch.from_numpy(d).to(dev), torch.from_numpy(p).to(dev), 1, 1, 0, np.float32(0.4), return_stats=True)
    keep = keep.cpu().numpy()
    assert np.array_equal(keep, ref_keep), "mismatching candidates: %s (pairs %d, general path %d)" % (np.flatnonzero(keep != ref_keep)[:10], stats[0], stats[1])


@pytest.mark.parametrize("defer_exact", [None, 0, 1, 3])
def test_nms3d_256_keep_array_equals_reference_golden(defer_exact):
    """config 3 (S3D-nuclei 256^3, 150 606 candidates, Rays_GoldenSpiral(96)): survivors bit-identical to the compiled reference
    run with ONE OpenMP thread (minutes of Qhull, hence the committed golden bits) -- with the library's defaults and with the exact
    volumes carried into the tail batch from round 1 / round 3 on / never ("nms3d_defer_exact")"""
    import contextlib
    from oracle import synth
    from stardist_amd.lib import _native as N, stardist3d as sd3
    with (contextlib.nullcontext() if defer_exact is None else N.option("nms3d_defer_exact", defer_exact)):
        _nms3d_256_golden(synth, sd3)


def _nms3d_256_golden(synth, sd3):
    from stardist_amd.rays3d import Rays_GoldenSpiral
    rays = Rays_GoldenSpiral(96)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    d, p, s, nobj = synth.s3d_nuclei(256, V)
    g = _golden()
    assert int(g["nms3d_256_n"]) == len(d)
    keep = sd3.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(0.3))
    ref_keep = np.unpackbits(g["nms3d_256_keep"])[:len(d)].astype(bool)
    assert int(ref_keep.sum()) == 1328
    assert np.array_equal(keep, ref_keep), "mismatching candidates: %s" % np.flatnonzero(keep != ref_keep)[:10]


@pytest.mark.parametrize("overlap_label,variant", [(None, "golden96"), (-1, "golden96"), (None, "golden48-aniso-grid122"), (None, "cartesian-grid211")])
def test_predict_instances_3d_end_to_end_equals_oracle_composition(refmods, overlap_label, variant):
    """StarDist3D.predict_instances (dense path, so that both sides see ONE forward pass) vs the reference composition on the same
    prob/dist maps: _ind_prob_thresh -> sort -> c_non_max_suppression_inds (1 thread) -> c_polyhedron_to_label -> relabel_sequential
    with the negative-overlap-label remapping of model3d.py:634-645"""
    import torch
    import bench
    from oracle import port, synth
    from stardist_amd import nms
    from stardist_amd.models import Config3D, StarDist3D
    dev = torch.device("cuda:0")
    vol = synth.s3d_nuclei_image(64, seed=3)
    import warnings
    from stardist_amd.rays3d import Rays_Cartesian, Rays_GoldenSpiral
    if variant == "golden96":
        cfg = Config3D(rays=96)
    elif variant == "golden48-aniso-grid122":            # (round 6) anisotropic rays on an anisotropic grid
        cfg = Config3D(rays=Rays_GoldenSpiral(48, anisotropy=(2, 1, 1)), grid=(1, 2, 2), anisotropy=(2, 1, 1))
    else:                                                # (round 6) Rays_Cartesian: degenerate pole faces
        cfg = Config3D(rays=Rays_Cartesian(8, 5), grid=(2, 1, 1))
    grid3 = tuple(cfg.grid)
    warnings.filterwarnings("ignore", message=".*coincide in float32.*")
    model = StarDist3D(cfg, basedir=None, device=dev, seed=0)
    model.thresholds = dict(prob=0.5, nms=0.3)
    bench.calibrate_heads(model, torch.from_numpy(vol).to(dev), frac=0.02, radius=8.5, noise=0.03)
    (labels, res), (prob, dist) = model.predict_instances(vol, return_predict=True, overlap_label=overlap_label)
    prob = np.asarray(prob); dist = np.asarray(dist)
    rays = res["rays"]
    V, F = np.asarray(rays.vertices, np.float32), np.asarray(rays.faces, np.int32)
    # ---- reference composition (stardist/nms.py:233-282, model3d.py:589-674)
    mask = port.ind_prob_thresh(prob, 0.5, b=2)
    pts = np.stack(np.where(mask), 1)
    pr = prob[mask]; di = dist[mask]
    order = nms._argsort_desc(pr)                       # ties: stable order on both sides (DESIGN.md deviation 6)
    pr, di, pts = pr[order], di[order], pts[order]
    pts = pts * np.array(grid3).reshape(1, 3)
    refmods.stardist3d(); refmods.set_threads(1)
    keep = refmods.stardist3d().c_non_max_suppression_inds(np.ascontiguousarray(di, np.float32), np.ascontiguousarray(pts, np.float32), V, F,
                                                           np.ascontiguousarray(pr, np.float32), 1, 1, 0, np.float32(0.3))
    keep = keep.astype(bool)
    assert 10 < keep.sum() < len(keep)
    pts_s, pr_s, di_s = pts[keep], pr[keep], di[keep]
    lab_ref = port.polyhedron_to_label(di_s, pts_s, V, F, vol.shape, prob=pr_s, overlap_label=overlap_label)
    if overlap_label is not None and overlap_label < 0 and (overlap_label in lab_ref):
        m = lab_ref == overlap_label
        ol2 = max(set(np.unique(lab_ref)) - {overlap_label}) + 1
        lab_ref[m] = ol2
        lab_ref, fwd, bwd = port.relabel_sequential(lab_ref)
        lab_ref[lab_ref == fwd[ol2]] = overlap_label
    else:
        lab_ref, _, _ = port.relabel_sequential(lab_ref)
    assert np.array_equal(res["points"], pts_s) and np.array_equal(res["prob"], pr_s) and np.array_equal(res["dist"], di_s)
    assert labels.shape == lab_ref.shape and np.array_equal(labels, lab_ref)
    if overlap_label is not None:
        assert (labels == overlap_label).any()           # the synthetic spheres do overlap: the branch is exercised


@pytest.mark.parametrize("case", ["2D_demo-fixture", "default-synthetic", "rays64-grid12", "rays128-grid21"])
def test_predict_instances_2d_end_to_end_equals_oracle_composition(refmods, case, monkeypatch, tmp_path):
    """BASELINE config 1's substitute: the reference's `models/examples/2D_demo` topology (config.json: grid (2,2),
    thresholds.json) with seeded weights on the reference's own fixture image tests/data/img2d.tif (stored in
    tests/golden/fixture_images.npz), normalised as tests/test_model2D.py:19 does -- and the default (grid 1) model on a synthetic
    tile.  StarDist2D.predict_instances (dense path: both sides see ONE forward pass) vs the reference composition on the same
    prob/dist maps: _ind_prob_thresh -> sort -> points * grid -> compiled c_non_max_suppression_inds -> polygons_to_label
    (stardist/nms.py:76-132, model2d.py:512-563, geom2d.py:130-197).  Label image and result dict must be identical."""
    import shutil
    import torch
    import bench
    from oracle import port, synth
    from stardist_amd import nms
    from stardist_amd.models import Config2D, StarDist2D
    from stardist_amd.utils import normalize
    dev = torch.device("cuda:0")
    if case == "2D_demo-fixture":
        cache = tmp_path / "cache"
        shutil.copytree(os.path.join(ROOT, "tests", "golden", "pretrained"), str(cache))
        monkeypatch.setenv("STARDIST_AMD_MODELS", str(cache))
        model = StarDist2D.from_pretrained("2D_demo", device=dev)
        assert tuple(model.config.grid) == (2, 2) and abs(model.thresholds.prob - 0.4861655269131771) < 1e-12 and model.thresholds.nms == 0.5
        img = normalize(np.load(os.path.join(ROOT, "tests", "golden", "fixture_images.npz"))["img2d"])
        assert img.shape == (256, 256) and img.dtype == np.float32
        bench.calibrate_heads(model, torch.from_numpy(img).to(dev), frac=0.08, radius=7.0)
    elif case.startswith("rays"):
        # (round 6) more than 32 rays and an anisotropic grid: the NMS takes the sweeps of the larger vertex capacities and their general paths
        R, g = (64, (1, 2)) if case == "rays64-grid12" else (128, (2, 1))
        model = StarDist2D(Config2D(n_rays=R, grid=g), basedir=None, device=dev, seed=0)
        img = synth.s2d_nuclei_image(192, 224, seed=5)
        bench.calibrate_heads(model, torch.from_numpy(img).to(dev))
    else:
        model = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
        img = synth.s2d_nuclei_image(384, 512, seed=4)
        bench.calibrate_heads(model, torch.from_numpy(img).to(dev))
    pt, nt, grid = model.thresholds.prob, model.thresholds.nms, tuple(model.config.grid)
    (labels, res), (prob, dist) = model.predict_instances(img, return_predict=True)
    prob = np.asarray(prob); dist = np.asarray(dist)
    assert prob.shape == tuple(s // g for s, g in zip(img.shape, grid)) and dist.shape == prob.shape + (model.config.n_rays,)
    # ---- reference composition
    mask = port.ind_prob_thresh(prob, pt, b=2)
    pts = np.stack(np.where(mask), 1)
    di, sc = dist[mask], prob[mask]
    order = nms._argsort_desc(sc)                       # ties: stable order on both sides (DESIGN.md deviation 6)
    di, sc, pts = di[order], sc[order], pts[order]
    pts = pts * np.array(grid).reshape(1, 2)
    refmods.set_threads(min(os.cpu_count() or 1, 16))
    keep = refmods.stardist2d().c_non_max_suppression_inds(np.ascontiguousarray(di, np.float32), np.ascontiguousarray(pts.astype(np.int32).astype(np.float32)),
                                                           1, 1, 0, np.float32(nt)).astype(bool)
    assert 10 < keep.sum() < len(keep)
    lab_ref = port.polygons_to_label(di[keep], pts[keep], prob=sc[keep], shape=img.shape)
    coord_ref = port.dist_to_coord(di[keep], pts[keep])
    assert np.array_equal(res["points"], pts[keep]) and np.array_equal(res["prob"], sc[keep])
    assert res["coord"].dtype == np.float32 and np.array_equal(res["coord"], coord_ref)
    assert labels.shape == lab_ref.shape and labels.dtype == lab_ref.dtype and np.array_equal(labels, lab_ref)
    assert labels.max() == keep.sum()
    # the sparse path (default of predict_instances) gives the same instances
    labels_s, res_s = model.predict_instances(img)
    assert np.array_equal(labels_s, labels) and np.array_equal(res_s["coord"], res["coord"]) and np.array_equal(res_s["prob"], res["prob"])
