"""GPU: BASELINE.json full-size workloads checked through size-independent properties and the calibration counts the
survey measured with the compiled reference (SURVEY.md section 8d generator specification)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_nms2d_2048_count_idempotence_and_order():
    """S2D-uniform 2048^2: 416 700 candidates -> 25 628 survivors with the reference;
    NMS of the survivors keeps all of them (idempotence); survivors are pairwise below the threshold by construction."""
    from oracle import synth
    from stardist_amd.lib import stardist2d as sd2
    d, p, s = synth.s2d_uniform(2048, 2048)
    assert len(d) == 416700
    keep = sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4))
    assert int(keep.sum()) == 25628
    assert keep[0]                                   # the best candidate always survives
    keep2 = sd2.c_non_max_suppression_inds(d[keep], p[keep], 1, 1, 0, np.float32(0.4))
    assert keep2.all()
    # appending suppressed candidates at the end (lowest rank) cannot change who survives among the first ones
    sup = np.flatnonzero(~keep)[:5000]
    d3 = np.concatenate([d[keep], d[sup]]); p3 = np.concatenate([p[keep], p[sup]])
    keep3 = sd2.c_non_max_suppression_inds(d3, p3, 1, 1, 0, np.float32(0.4))
    assert keep3[:int(keep.sum())].all() and not keep3[int(keep.sum()):].any()


def test_nms2d_1024_bit_exact_vs_reference(refmods):
    from oracle import synth
    from stardist_amd.lib import stardist2d as sd2
    d, p, s = synth.s2d_uniform(1024, 1024)
    ref_keep = refmods.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4))
    keep = sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4))
    assert (len(d), int(ref_keep.sum())) == (104580, 6439)
    assert np.array_equal(keep, ref_keep)


@pytest.mark.parametrize("R", [64, 100])
def test_nms2d_many_rays_vs_reference(refmods, R):
    """n_rays > 32 goes through the larger-capacity kernel instantiations"""
    from oracle import synth
    from stardist_amd.lib import stardist2d as sd2
    d, p, s = synth.s2d_uniform(200, 220, n_rays=R, radius=14, noise=0.3)
    ref_keep = refmods.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4))
    keep = sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4))
    assert np.array_equal(keep, ref_keep)


def test_raster2d_label_equals_polygon_mask():
    """the reference's tests/test_big.py:202-213 property: with non-overlapping painting order, label i is exactly the
    mask of polygon i where no later polygon covers it; checked via the oracle port on the NMS survivors of a 512^2 tile"""
    from oracle import port, synth
    from stardist_amd.lib import stardist2d as sd2
    d, p, s = synth.s2d_uniform(512, 512)
    keep = sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4))
    d, p, s = d[keep], p[keep], s[keep]
    coord = port.dist_to_coord(d, p)
    ind = np.argsort(s, kind="stable")
    lbl = sd2.c_polygons_to_label(coord[ind], ind.astype(np.int32), (512, 512))
    assert np.array_equal(lbl, port.polygons_to_label(d, p, (512, 512), prob=s))
    top = int(np.argmax(s))                          # the best polygon is painted last: its label is its full mask
    rr, cc = port.polygon(coord[top, 0], coord[top, 1], (512, 512))
    m = np.zeros((512, 512), bool); m[rr, cc] = True
    assert np.array_equal(lbl == top + 1, m)


def test_nms3d_256_calibration_count_and_raster():
    """S3D-nuclei 256^3: 150 606 candidates -> 1 328 survivors with the reference"""
    from oracle import synth
    from stardist_amd.lib import stardist3d as sd3
    from stardist_amd.rays3d import Rays_GoldenSpiral
    rays = Rays_GoldenSpiral(96)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    d, p, s, nobj = synth.s3d_nuclei(256, V)
    assert (len(d), nobj) == (150606, 1331)
    keep = sd3.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(0.3))
    assert int(keep.sum()) == 1328
    keep2 = sd3.c_non_max_suppression_inds(d[keep], p[keep], V, F, s[keep], 1, 1, 0, np.float32(0.3))
    assert keep2.all()
    lbl = sd3.c_polyhedron_to_label(d[keep], p[keep], V, F, np.arange(1, keep.sum() + 1, dtype=np.int32), 0, 0, 0, 0, (256, 256, 256))
    assert lbl.shape == (256, 256, 256) and lbl.max() == keep.sum()
    # every surviving centre is labelled with its own id unless an earlier (higher-scored) polyhedron covers it
    own = lbl[tuple(p[keep].astype(int).T)]
    assert (own > 0).all() and (own <= np.arange(1, keep.sum() + 1)).all()


def test_predict_instances_dense_equals_sparse_and_big_equals_whole():
    """reference properties tests/test_model2D.py:442-450 (dense == sparse) and tests/test_big.py:86-117 (big == whole image:
    same objects), on the network's own output"""
    import torch
    import bench
    from oracle import synth
    from stardist_amd.models import Config2D, StarDist2D
    dev = torch.device("cuda:0")
    img = synth.s2d_nuclei_image(512, 512, seed=5)
    model = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    bench.calibrate_heads(model, torch.from_numpy(img).to(dev), frac=0.03)
    l1, r1 = model.predict_instances(img, sparse=True)
    l2, r2 = model.predict_instances(img, sparse=False)
    # two forward passes: the network is run-to-run deterministic (small deep layers as one GEMM, models/unet.py), so both agree exactly
    assert np.array_equal(r1["points"], r2["points"]) and np.array_equal(r1["coord"], r2["coord"])
    assert np.array_equal(l1, l2)
    l3, r3 = model.predict_instances(img, n_tiles=(2, 2))
    assert len(r3["prob"]) == len(r1["prob"]) and (l3 > 0).sum() == pytest.approx((l1 > 0).sum(), rel=1e-3)
    lb, rb = model.predict_instances_big(img, axes="YX", block_size=256, min_overlap=64, context=64, show_progress=False)
    assert lb.shape == l1.shape and len(rb["prob"]) == len(r1["prob"])
    # same objects (tests/test_big.py:98-117 asks for matching accuracy 1 and polys allclose(atol=1e-2), i.e. not bit equality):
    # every whole-image centre has a block-wise centre within one pixel, foreground masks agree up to border pixels
    a = np.asarray(r1["points"], np.float64); b = np.asarray(rb["points"], np.float64)
    d = np.sqrt(((a[:, None] - b[None]) ** 2).sum(-1)).min(1)
    assert (d <= 1.5).mean() >= 0.99
    assert np.count_nonzero((lb > 0) != (l1 > 0)) <= 1e-3 * l1.size


def test_predict_instances_sharded_single_rank_equals_whole_image():
    """design A (block-wise local NMS + final cross-tile NMS; here one rank, no process group) on the network's own output:
    same objects as predict_instances on the whole image (up to float noise between differently shaped forward passes)"""
    import torch
    import bench
    from oracle import synth
    from stardist_amd.models import Config2D, StarDist2D
    dev = torch.device("cuda:0")
    img = synth.s2d_nuclei_image(512, 512, seed=7)
    model = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    bench.calibrate_heads(model, torch.from_numpy(img).to(dev), frac=0.03)
    l1, r1 = model.predict_instances(img)
    ls, rs = model.predict_instances_sharded(img, "YX", block_size=256, min_overlap=64, context=64)
    assert ls.shape == l1.shape and abs(len(rs["prob"]) - len(r1["prob"])) <= max(2, 0.01 * len(r1["prob"]))
    a = np.asarray(r1["points"], np.float64); b = np.asarray(rs["points"], np.float64)
    d = np.sqrt(((a[:, None] - b[None]) ** 2).sum(-1)).min(1)
    assert (d <= 1.5).mean() >= 0.98
    assert np.count_nonzero((ls > 0) != (l1 > 0)) <= 2e-3 * l1.size


@pytest.mark.parametrize("dim", ["2d", "3d", "2d-multiclass"])
def test_sharded_window_tiles_equal_whole_image_raster_and_band_nms_equals_union_nms(dim):
    """design A on the device: (1) the tiles every rank renders for the write regions of its blocks (windowed rasterisers, 3D incl. the
    global relabel_sequential) are the corresponding parts of the label image the whole-image rasteriser produces from the same
    final instances; (2) restricting the cross-tile NMS to the band survivors gives the instances of an NMS over the whole union
    (the round-2 formulation); (3) multi-class: class probabilities travel with the survivors."""
    import torch
    import bench
    from oracle import synth
    from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D
    dev = torch.device("cuda:0")
    if dim == "3d":
        img = synth.s3d_nuclei_image(96, seed=5)
        model = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
        model.thresholds = dict(prob=0.5, nms=0.3)
        bench.calibrate_heads(model, torch.from_numpy(img).to(dev), frac=0.02, radius=8.5, noise=0.03)
        args = dict(axes="ZYX", block_size=64, min_overlap=16, context=8)
    else:
        img = synth.s2d_nuclei_image(768, 1024, seed=5)
        model = StarDist2D(Config2D(n_rays=32, n_classes=(3 if dim == "2d-multiclass" else None)), basedir=None, device=dev, seed=0)
        bench.calibrate_heads(model, torch.from_numpy(img).to(dev), frac=0.05)
        args = dict(axes="YX", block_size=384, min_overlap=64, context=64)
    labels, res = model.predict_instances_sharded(img, **args)
    st = dict(model._last_sharded_stats)
    assert st["band"] + st["interior"] == st["unique"] and st["interior"] > 0 and st["band"] > 0 and st["instances"] == len(res["prob"]) > 20
    tiles, res2 = model.predict_instances_sharded(img, labels_out="local", **args)
    assert np.array_equal(res2["points"], res["points"]) and len(tiles) == st["blocks"]
    for bi, sl, t in tiles:
        assert np.array_equal(t.cpu().numpy(), labels[sl]), bi
    assert len(st["per_block"]) == st["blocks"]
    if dim == "2d-multiclass":
        assert res["class_prob"].shape == (len(res["prob"]), 4) and np.allclose(res["class_prob"].sum(1), 1, atol=1e-5)
        assert np.array_equal(res["class_id"], res["class_prob"].argmax(1))
    # (2): all survivors through the final NMS -- force every survivor into the band by a huge margin
    import stardist_amd.big as B
    orig = B._exclusive_intervals
    B._exclusive_intervals = lambda blocks, axes_out: np.stack([np.full((len(blocks), len(axes_out)), np.inf), np.full((len(blocks), len(axes_out)), -np.inf)], -1)
    try:
        labels_u, res_u = model.predict_instances_sharded(img, **args)
        assert model._last_sharded_stats["interior"] == 0
    finally:
        B._exclusive_intervals = orig
    assert np.array_equal(res_u["points"], res["points"]) and np.array_equal(res_u["prob"], res["prob"]) and np.array_equal(labels_u, labels)


@pytest.mark.parametrize("dim", ["2d", "3d"])
def test_sharded_prediction_without_detections(dim):
    """an input on which nothing passes the probability threshold: every form of the label output is background of the right shape
    (the windowed 3D rasteriser used to return a host array of the WHOLE volume on its empty paths)"""
    import torch
    from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D
    dev = torch.device("cuda:0")
    if dim == "3d":
        model = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
        img = np.zeros((64, 64, 96), np.float32)
        args = dict(axes="ZYX", block_size=48, min_overlap=16, context=8)
    else:
        model = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
        img = np.zeros((256, 384), np.float32)
        args = dict(axes="YX", block_size=128, min_overlap=32, context=32)
    with torch.no_grad():
        model.net.prob.bias.fill_(-20.0)                 # sigmoid(-20): nothing above any threshold
    labels, res = model.predict_instances_sharded(img, **args)
    assert labels.shape == img.shape and not np.asarray(labels).any() and len(res["prob"]) == 0
    tiles, res2 = model.predict_instances_sharded(img, labels_out="local", **args)
    assert len(res2["prob"]) == 0 and len(tiles) == model._last_sharded_stats["blocks"]
    for bi, sl, t in tiles:
        assert torch.is_tensor(t) and tuple(t.shape) == tuple(s.stop - s.start for s in sl) and not bool(t.any())
