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
