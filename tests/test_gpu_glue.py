"""GPU: the glue natives and host paths added in round 5, each against the numpy statement of the reference line it replaces:
sd_sorted_rows_device (score order -> feature rows + pixel coordinates), sd_dist_to_coord_device (geom2d.py:130-146 in numpy's
arithmetic), predict_instances_iter (uploads overlapped; same results as predict_instances per image), to_host_many."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _vp(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else None)


@pytest.mark.parametrize("nd,full,grid", [(2, (37, 91), (1, 1)), (2, (64, 50), (2, 4)), (3, (9, 20, 33), (1, 2, 2))])
def test_sorted_rows_matches_numpy(nd, full, grid):
    import torch
    from stardist_amd.lib import _native as N
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(nd * 7 + full[0])
    n_sel = 4000
    pts = np.stack([rng.randint(0, s, n_sel) for s in full], 1).astype(np.int32)
    n = 3111
    order = rng.permutation(n_sel)[:n].astype(np.int64)
    origin = np.zeros(nd, np.int32)
    tp, to = torch.from_numpy(pts).to(dev), torch.from_numpy(order).to(dev)
    rows = torch.full((n,), -1, dtype=torch.int64, device=dev)
    pf = torch.full((n, nd), float("nan"), dtype=torch.float32, device=dev)
    pi = torch.full((n, nd), -1, dtype=torch.int64, device=dev)
    N.dcall(tp, "sd_sorted_rows_device", _vp(tp), _vp(to), n, nd, N.ptr(np.asarray(full, np.int32)), N.ptr(origin), N.ptr(np.asarray(grid, np.int32)),
            _vp(rows), _vp(pf), _vp(pi))
    sel = pts[order].astype(np.int64)
    assert np.array_equal(rows.cpu().numpy(), np.ravel_multi_index(tuple(sel.T), full))
    assert np.array_equal(pi.cpu().numpy(), sel * np.array(grid))
    assert np.array_equal(pf.cpu().numpy(), (sel * np.array(grid)).astype(np.float32))


@pytest.mark.parametrize("n,levels", [(1, 0), (2, 1), (257, 0), (5000, 7), (418577, 0), (418577, 300), (1 << 20, 2)])
def test_sort_scores_desc_equals_numpy_stable_argsort_reversed(n, levels):
    """sd_sort_scores_desc_device == np.argsort(scores, kind="stable")[::-1] (stardist/nms.py:114,167 as nms._argsort_desc states it), with many
    equal scores (levels > 0: that many distinct values) -- and nms._sort_desc / _argsort_desc take that route for float32 GPU scores"""
    import torch
    from stardist_amd.lib import _native as N
    from stardist_amd.nms import _argsort_desc, _sort_desc
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(n % 1000 + levels)
    x = rng.rand(n).astype(np.float32)
    if levels:
        x = (np.floor(x * levels) / levels).astype(np.float32)
    x[rng.randint(0, n, max(1, n // 50))] = np.float32(1.0)
    ref = np.argsort(x, kind="stable")[::-1]
    t = torch.from_numpy(x).to(dev)
    sp = torch.full((n,), float("nan"), dtype=torch.float32, device=dev)
    order = torch.full((n,), -1, dtype=torch.int64, device=dev)
    N.dcall(t, "sd_sort_scores_desc_device", _vp(t), n, _vp(sp), _vp(order))
    assert np.array_equal(order.cpu().numpy(), ref)
    assert np.array_equal(sp.cpu().numpy(), x[ref])
    sp2, order2 = _sort_desc(t)
    assert np.array_equal(order2.cpu().numpy(), ref) and np.array_equal(sp2.cpu().numpy(), x[ref])
    assert np.array_equal(_argsort_desc(t).cpu().numpy(), ref)
    # the framework route (any other dtype) states the same order
    assert np.array_equal(_argsort_desc(t.double()).cpu().numpy(), ref)


@pytest.mark.parametrize("R,scale", [(32, (1, 1)), (11, (1, 1)), (32, (0.5, 2.0)), (64, (1.25, 1))])
def test_dist_to_coord_native_equals_numpy(R, scale):
    """sd_dist_to_coord_device == the reference's numpy expression bit for bit (float32 x float64 products rounded to float32, optional
    scale, centre added in float64 and rounded once), integer and float centres"""
    import torch
    from oracle import port
    from stardist_amd.geometry.geom2d import dist_to_coord
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(R)
    n = 5003
    dist = (rng.uniform(0.001, 300, (n, R))).astype(np.float32)
    for pts in (rng.randint(0, 16384, (n, 2)).astype(np.int64), rng.uniform(0, 9000, (n, 2))):
        want = port.dist_to_coord(dist, pts, scale_dist=scale)
        got = dist_to_coord(torch.from_numpy(dist).to(dev), torch.from_numpy(pts).to(dev), scale_dist=scale)
        assert got.dtype == torch.float32 and tuple(got.shape) == (n, 2, R)
        assert np.array_equal(got.cpu().numpy(), want.astype(np.float32))
    assert tuple(dist_to_coord(torch.zeros((0, R), device=dev), torch.zeros((0, 2), dtype=torch.int64, device=dev)).shape) == (0, 2, R)


@pytest.mark.parametrize("dim", ["2d", "3d"])
def test_predict_instances_iter_equals_predict_instances(dim):
    """a stream of host arrays through predict_instances_iter (helper thread, page-locked staging ring, copy stream): per image the
    labels and the dict of predict_instances, in order -- different images, a repeated one, and a device tensor passed through"""
    import torch
    import bench
    from oracle import synth
    from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D
    dev = torch.device("cuda:0")
    if dim == "2d":
        imgs = [synth.s2d_nuclei_image(256, 320, seed=s) for s in (1, 2, 3)]
        m = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
        bench.calibrate_heads(m, torch.from_numpy(imgs[0]).to(dev))
    else:
        imgs = [synth.s3d_nuclei_image(64, seed=s) for s in (1, 2)]
        m = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
        m.thresholds = dict(prob=0.5, nms=0.3)
        bench.calibrate_heads(m, torch.from_numpy(imgs[0]).to(dev), frac=0.02, radius=8.5, noise=0.03)
    stream = imgs + [imgs[0], torch.from_numpy(imgs[1]).to(dev)]
    want = [m.predict_instances(x) for x in stream]
    got = list(m.predict_instances_iter(iter(stream), prefetch=2))
    assert len(got) == len(want)
    for (lw, rw), (lg, rg) in zip(want, got):
        assert np.array_equal(lw, lg)
        for k in ("points", "prob"):
            assert np.array_equal(rw[k], rg[k])
    assert len(want[0][1]["prob"]) > 5
    # an exception inside the consumer's step leaves no thread behind; a normaliser falls back to the plain loop
    gen = m.predict_instances_iter(iter(imgs))
    next(gen); gen.close()


def test_to_host_many_one_sync():
    import torch
    from stardist_amd.utils import to_host_many
    dev = torch.device("cuda:0")
    a = torch.arange(1000, device=dev, dtype=torch.int32).reshape(10, 100)
    b = torch.rand(7, 3, device=dev)
    out = to_host_many([a, None, b, torch.ones(3)])
    assert np.array_equal(out[0], a.cpu().numpy()) and out[1] is None and np.array_equal(out[2], b.cpu().numpy()) and np.array_equal(out[3], np.ones(3, np.float32))


@pytest.mark.parametrize("n,R,frac,ties", [(5000, 32, 0.1, False), (5000, 32, 0.3, True), (3000, 11, 0.02, True), (4097, 64, 0.5, False), (7, 32, 1.0, True), (1500, 32, 0.0, False)])
def test_survivors_of_sorted_equals_the_reference_lines(n, R, frac, ties):
    """csrc/survivors.hip against the numpy statement of model2d.py:536-561 + geom2d.py:130-146, 186-197: positions of the keep flags, the
    survivors' rows, dist_to_coord, and the painting order `np.argsort(prob, kind='stable')` with label ids = position in NMS order (+ 1
    in the rasteriser) -- with TIES in the scores (groups of equal probabilities keep their NMS order while the groups reverse)"""
    import torch
    from stardist_amd.lib.stardist2d import survivors_of_sorted
    from stardist_amd.geometry.geom2d import ray_angles
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(n + R)
    prob = np.sort(rng.uniform(0.3, 1, n).astype(np.float32))[::-1].copy()
    if ties:
        prob = np.sort(np.round(prob, 2))[::-1].copy()             # long runs of equal scores
    pts = rng.randint(0, 3000, (n, 2)).astype(np.int64)
    dist = rng.uniform(0.5, 40, (n, R)).astype(np.float32)
    keep = rng.uniform(0, 1, n) < frac
    got = survivors_of_sorted(torch.from_numpy(keep).to(dev), torch.from_numpy(prob).to(dev), torch.from_numpy(pts).to(dev), torch.from_numpy(dist).to(dev))
    oprob, opts, coord, cpaint, lpaint = [g.cpu().numpy() for g in got]
    idx = np.flatnonzero(keep)
    assert np.array_equal(oprob, prob[idx]) and np.array_equal(opts, pts[idx])
    phis = ray_angles(R)
    want = (dist[idx][:, np.newaxis] * np.array([np.sin(phis), np.cos(phis)])).astype(np.float32)
    want += pts[idx][..., np.newaxis]
    assert coord.dtype == np.float32 and np.array_equal(coord, want)
    ind = np.argsort(prob[idx], kind="stable")
    assert np.array_equal(lpaint, ind.astype(np.int32)) and np.array_equal(cpaint, want[ind])


def test_predict_instances_2d_fused_survivors_equal_generic_path():
    """predict_instances through the fused survivors path == the generic chain (positions -> gathers -> dist_to_coord -> stable sort ->
    rasteriser): labels and every entry of the result dict, on a prediction with a few thousand instances"""
    import torch
    import bench
    from oracle import synth
    from stardist_amd.models import Config2D, StarDist2D
    dev = torch.device("cuda:0")
    img = torch.from_numpy(synth.s2d_nuclei_image(768, 640, seed=3)).to(dev)
    m = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    bench.calibrate_heads(m, img)
    lab1, res1 = m.predict_instances(img)
    # the generic chain, called directly on the same candidates
    from stardist_amd.nms import non_maximum_suppression_sparse_sorted
    cand = None
    for r in m._predict_sparse_generator(img, axes=None, _presort=True):
        cand = r if r is not None else cand
    idx = non_maximum_suppression_sparse_sorted(cand.dist, cand.prob, cand.points_f32, nms_thresh=m.thresholds.nms)
    lab2, res2 = m._instances_from_survivors(tuple(img.shape), cand.points.index_select(0, idx), cand.prob.index_select(0, idx), cand.dist.index_select(0, idx))
    assert len(res1["prob"]) > 300
    assert np.array_equal(lab1, lab2)
    for k in ("coord", "points", "prob"):
        assert res1[k].dtype == res2[k].dtype and np.array_equal(res1[k], res2[k]), k
