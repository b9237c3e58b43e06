"""GPU parity tests (3D): HIP path through the C ABI vs the compiled reference (oracle/_ref)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rays(n=96, anisotropy=None):
    from stardist_amd.rays3d import Rays_GoldenSpiral
    return Rays_GoldenSpiral(n, anisotropy=anisotropy)


def _random_candidates(shape, n_rays, noise, seed, prob_thresh=0.9, radius=10):
    """after the reference's tests/test_nms3D.py:8-14 (create_random_data), seeded"""
    rng = np.random.RandomState(seed)
    dist = radius * np.ones(shape + (n_rays,))
    dist *= (1 + np.clip(noise, 0, 1) * rng.uniform(-1, 1, dist.shape))
    prob = rng.uniform(0, 1, shape)
    mask = prob > prob_thresh
    m2 = np.zeros_like(mask); m2[2:-2, 2:-2, 2:-2] = True
    mask &= m2
    pts = np.stack(np.where(mask), 1)
    d = dist[mask].astype(np.float32); s = prob[mask].astype(np.float32)
    ind = np.argsort(s)[::-1]
    return np.ascontiguousarray(d[ind]), np.ascontiguousarray(pts[ind].astype(np.float32)), np.ascontiguousarray(s[ind])


@pytest.mark.parametrize("n_rays,grid", [(32, (1, 1, 1)), (96, (1, 2, 2)), (17, (2, 1, 4))])
def test_star_dist3d_bit_exact(refmods, n_rays, grid):
    from stardist_amd.lib import stardist3d as sd3
    rays = _rays(n_rays)
    rng = np.random.RandomState(1)
    lbl = np.zeros((40, 50, 46), np.uint16)
    for k in range(12):
        c = rng.uniform(8, 36, 3); r = rng.uniform(4, 9)
        zz, yy, xx = np.mgrid[:40, :50, :46]
        m = ((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) <= r * r
        lbl[m & (lbl == 0)] = k + 1
    dz, dy, dx = (np.ascontiguousarray(v, np.float32) for v in rays.vertices.T)
    ref_d = refmods.stardist3d().c_star_dist3d(lbl, dz, dy, dx, n_rays, *grid)
    d = sd3.c_star_dist3d(lbl, dz, dy, dx, n_rays, *grid)
    assert d.shape == ref_d.shape
    assert np.array_equal(d, ref_d), np.abs(d - ref_d).max()


_REF_KEEP = {}


def _ref_keep_random(refmods, shape, n_rays, noise, thr):
    """the compiled reference's survivors on a seeded random candidate set; the single-threaded Qhull run takes up to 35 s, and two
    tests look at the same sets -- computed once per process"""
    key = (tuple(shape), n_rays, noise, float(thr))
    if key not in _REF_KEEP:
        rays = _rays(n_rays)
        d, p, s = _random_candidates(shape, n_rays, noise, seed=n_rays)
        _REF_KEEP[key] = refmods.stardist3d().c_non_max_suppression_inds(d, p, rays.vertices, rays.faces.astype(np.int32), s, 1, 1, 0, np.float32(thr))
    return _REF_KEEP[key]


@pytest.mark.parametrize("mode,overlap", [(0, None), (1, None), (2, None), (3, None), (4, None), (0, 77), (0, -3), (2, 5)])
def test_polyhedron_to_label_identical(refmods, mode, overlap):
    """all five render modes of stardist3d_impl.cpp:1469-1509: full, kernel, hull (convex), bbox, debug"""
    from oracle import synth
    from stardist_amd.lib import stardist3d as sd3
    rays = _rays(96)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    d, p, s, _ = synth.s3d_nuclei(64, V, rc=1)
    rng = np.random.RandomState(0)
    sel = rng.choice(len(d), 60, replace=False)
    d = (d[sel] * (1 + 0.2 * rng.uniform(-1, 1, d[sel].shape))).astype(np.float32); p = p[sel]
    labels = np.arange(1, len(d) + 1, dtype=np.int32)
    args = (d, p, V, F, labels, mode, 0, int(overlap is not None), int(0 if overlap is None else overlap), (64, 64, 64))
    ref_lbl = refmods.stardist3d().c_polyhedron_to_label(*args)
    lbl = sd3.c_polyhedron_to_label(*args)
    assert lbl.dtype == np.int32 and lbl.shape == ref_lbl.shape
    assert np.array_equal(lbl, ref_lbl), np.count_nonzero(lbl != ref_lbl)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("rays_name,radius,noise,vol", [("tetra1", 6, 0.3, 36), ("octo1", 5, 0.5, 36), ("cartesian", 6, 0.3, 36), ("golden12", 7, 0.9, 40),
                                                        ("golden32", 0.7, 0.3, 24), ("golden32", 22, 0.2, 40), ("golden96", 9, 0.6, 48)])
def test_polyhedron_to_label_extreme_shapes(refmods, rays_name, radius, noise, vol, mode):
    """the rasteriser on what the random nuclei never are: ray sets of 4 / 6 rays, `Rays_Cartesian` (degenerate pole faces), very irregular
    polyhedra, polyhedra below one voxel, polyhedra larger than the volume (clipped on every side), centres on the border -- modes full,
    kernel, hull, bbox.  Mode full / hull: voxels exactly on a hull facet are exempt as everywhere."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _hull import on_hull_boundary
    from stardist_amd.lib import stardist3d as sd3
    from stardist_amd.rays3d import Rays_Cartesian, Rays_GoldenSpiral, Rays_Octo, Rays_Tetra
    import warnings
    rays = {"tetra1": lambda: Rays_Tetra(1), "octo1": lambda: Rays_Octo(1), "cartesian": lambda: Rays_Cartesian(8, 5), "golden12": lambda: Rays_GoldenSpiral(12),
            "golden32": lambda: Rays_GoldenSpiral(32), "golden96": lambda: Rays_GoldenSpiral(96)}[rays_name]()
    V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
    rng = np.random.RandomState(len(V) * 7 + int(radius * 10))
    n = 40
    d = (radius * (1 + noise * rng.uniform(-1, 1, (n, len(V))))).astype(np.float32)
    p = rng.uniform(0, vol - 1, (n, 3)).astype(np.float32)
    p[:6] = np.round(p[:6]); p[6] = (0, 0, 0); p[7] = (vol - 1, vol - 1, vol - 1); p[8] = (0, vol / 2, vol - 1)
    labels = np.arange(1, n + 1, dtype=np.int32)
    args = (d, p, V, F, labels, mode, 0, 0, 0, (vol, vol, vol))
    ref_lbl = refmods.stardist3d().c_polyhedron_to_label(*args)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lbl = sd3.c_polyhedron_to_label(*args)
    diff = np.argwhere(lbl != ref_lbl)
    if mode in (0, 2) and len(diff):
        assert on_hull_boundary(diff, p, d, V).all(), (rays_name, radius, mode, len(diff), diff[:6])
        assert len(diff) <= 8 * n
    else:
        assert len(diff) == 0, (rays_name, radius, mode, len(diff), diff[:6])
    assert (ref_lbl > 0).any() or mode == 1            # (the kernel of a polyhedron with degenerate faces can be empty)


@pytest.mark.parametrize("shape,n_rays,noise,thr", [((22, 33, 44), 32, 0.1, 0.2), ((22, 33, 44), 32, 0.5, 0.5),
                                                     ((22, 33, 44), 96, 0.3, 0.3), ((33, 44, 55), 14, 0.0, 0.2),
                                                     ((33, 44, 55), 22, 0.0, 0.4), ((22, 33, 44), 96, 0.3, 0.6)])
def test_nms3d_random_survivors(refmods, shape, n_rays, noise, thr):
    from stardist_amd.lib import stardist3d as sd3
    rays = _rays(n_rays)
    d, p, s = _random_candidates(shape, n_rays, noise, seed=n_rays)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    ref_keep = _ref_keep_random(refmods, shape, n_rays, noise, thr)
    keep, stats = sd3.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(thr), return_stats=True)
    diff = np.flatnonzero(keep != ref_keep)
    assert len(diff) == 0, "survivor mismatch at %s of %d (stats %s)" % (diff[:10], len(d), stats.tolist())


@pytest.mark.parametrize("nx,nz,shape,noise,thr", [(8, 5, (22, 33, 44), 0.3, 0.3), (11, 5, (22, 33, 44), 0.0, 0.2), (11, 5, (22, 33, 44), 0.5, 0.5),
                                                     (6, 4, (20, 30, 40), 0.2, 0.4)])
def test_nms3d_rays_cartesian_random_survivors(refmods, nx, nz, shape, noise, thr):
    """`Rays_Cartesian` (pole rays 1e-12 apart: degenerate faces, the reference's cascade on its error paths -- DESIGN.md section 4 item 3a) on
    RANDOM float candidates, as the reference's tests/test_nms3D.py draws them: same survivors (until round 6 only the closed ray sets
    were pinned; the lattice goldens of tests/test_gpu_lattice.py cover the integer case)"""
    import warnings
    from stardist_amd.lib import stardist3d as sd3
    from stardist_amd.rays3d import Rays_Cartesian
    rays = Rays_Cartesian(nx, nz)
    V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
    d, p, s = _random_candidates(shape, len(V), noise, seed=len(V))
    refmods.stardist3d(); refmods.set_threads(1)          # (the reference's anisotropy sum is only defined for one thread)
    ref_keep = refmods.stardist3d().c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(thr))
    import torch
    dev = torch.device("cuda:0")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        keep, stats = sd3.c_non_max_suppression_inds(*[torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (d, p, V, F, s)], 1, 1, 0, np.float32(thr), return_stats=True)
    keep = keep.cpu().numpy()                      # (device tensors in: the stats of the host-array entry point are not filled)
    diff = np.flatnonzero(keep != ref_keep)
    print("cartesian(%d, %d) %s noise %.1f thr %.1f: %d candidates -> %d survivors, kernel-stage suppressions %d, rendered pairs %d" %
          (nx, nz, shape, noise, thr, len(d), int(keep.sum()), int(stats[6]), int(stats[3])))
    assert int(stats[6]) == 0 and int(stats[2]) > 0 and int(stats[7]) > 0      # the kernel stage runs and never suppresses on such a mesh (Qhull error for every pair): the rendering decides
    assert len(diff) == 0, "survivor mismatch at %s of %d (stats %s)" % (diff[:10], len(d), stats.tolist())


@pytest.mark.parametrize("use_bbox,use_kdtree,thr", [(0, 1, 0.3), (1, 0, 0.3), (0, 0, 0.3), (0, 0, -0.5), (1, 1, -0.5), (1, 1, 0.0), (0, 1, 0.0)])
def test_nms3d_flag_combinations(refmods, use_bbox, use_kdtree, thr):
    """`use_bbox` / `use_kdtree` off and thresholds <= 0 (stardist3d_impl.cpp:1167-1175 every j > i instead of the radius search, :1221 the
    pretest only with use_bbox; a negative threshold suppresses on the first bound): same survivors as the compiled reference"""
    from stardist_amd.lib import stardist3d as sd3
    rays = _rays(32)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    d, p, s = _random_candidates((18, 24, 30), 32, 0.3, seed=5, prob_thresh=0.93, radius=6)
    assert 300 < len(d) < 800
    refmods.stardist3d(); refmods.set_threads(1)          # (the reference's anisotropy sum is only defined for one thread)
    ref_keep = refmods.stardist3d().c_non_max_suppression_inds(d, p, V, F, s, use_bbox, use_kdtree, 0, np.float32(thr))
    keep = sd3.c_non_max_suppression_inds(d, p, V, F, s, use_bbox, use_kdtree, 0, np.float32(thr))
    diff = np.flatnonzero(keep != ref_keep)
    assert len(diff) == 0, (use_bbox, use_kdtree, thr, diff[:10], int(ref_keep.sum()), int(keep.sum()))


@pytest.mark.parametrize("rays_name,shape,radius,noise,thr,pth", [("tetra1", (20, 24, 28), 5, 0.3, 0.3, 0.93), ("octo1", (20, 24, 28), 5, 0.4, 0.4, 0.93),
                                                                 ("golden32", (14, 16, 18), 0.7, 0.4, 0.3, 0.8), ("golden32", (16, 18, 20), 1.6, 0.6, 0.2, 0.85),
                                                                 ("golden32", (40, 44, 48), 16, 0.3, 0.3, 0.9985), ("golden12", (24, 26, 28), 6, 0.9, 0.5, 0.95)])
def test_nms3d_extreme_shapes(refmods, rays_name, shape, radius, noise, thr, pth):
    """the smallest ray sets (4 and 6 rays), polyhedra below one voxel (dist < 1: outside the cone map's preconditions) and polyhedra that fill
    a third of the volume (large rendering boxes), very irregular ones: same survivors as the compiled reference"""
    from stardist_amd.lib import stardist3d as sd3
    from stardist_amd.rays3d import Rays_GoldenSpiral, Rays_Octo, Rays_Tetra
    rays = {"tetra1": lambda: Rays_Tetra(1), "octo1": lambda: Rays_Octo(1), "golden32": lambda: Rays_GoldenSpiral(32), "golden12": lambda: Rays_GoldenSpiral(12)}[rays_name]()
    V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
    d, p, s = _random_candidates(shape, len(V), noise, seed=len(V) + int(radius * 10), prob_thresh=pth, radius=radius)
    assert 40 < len(d) < 3000, len(d)
    refmods.stardist3d(); refmods.set_threads(1)          # (the reference's anisotropy sum is only defined for one thread)
    ref_keep = refmods.stardist3d().c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(thr))
    keep = sd3.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(thr))
    diff = np.flatnonzero(keep != ref_keep)
    assert len(diff) == 0, (rays_name, radius, diff[:10], len(d), int(ref_keep.sum()), int(keep.sum()))


@pytest.mark.parametrize("n,thr,aniso", [(64, 0.3, None), (96, 0.3, None), (64, 0.5, (2, 1, 1))])
def test_nms3d_nuclei_survivors(refmods, n, thr, aniso):
    from oracle import synth
    from stardist_amd.lib import stardist3d as sd3
    rays = _rays(96, anisotropy=aniso)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    d, p, s, nobj = synth.s3d_nuclei(n, rays.vertices)
    ref_keep = refmods.stardist3d().c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(thr))
    keep, stats = sd3.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(thr), return_stats=True)
    assert np.array_equal(keep, ref_keep), (int(keep.sum()), int(ref_keep.sum()), stats.tolist())


def test_nms3d_volume_bounds_do_not_change_decisions(refmods, monkeypatch):
    """the inscribed / circumscribed polytope shortcut vs exact volumes for every pair: same survivors"""
    from oracle import synth
    from stardist_amd.lib import stardist3d as sd3
    rays = _rays(96)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    d, p, s, nobj = synth.s3d_nuclei(96, rays.vertices)
    keep_bounds = sd3.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(0.3))
    from stardist_amd.lib import _native as N
    with N.option("nms3d_volume_bounds", 0):
        keep_exact = sd3.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(0.3))
    ref_keep = refmods.stardist3d().c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(0.3))
    assert np.array_equal(keep_bounds, keep_exact) and np.array_equal(keep_exact, ref_keep)


def test_nms3d_tail_batch_does_not_change_survivors(refmods):
    """late greedy rounds as one speculative batch + device replay (nms3d.hip tail batch) vs plain rounds vs the reference"""
    import torch
    from oracle import synth
    from stardist_amd.lib import _native as N, stardist3d as sd3
    rays = _rays(96)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    d, p, s, nobj = synth.s3d_nuclei(96, rays.vertices)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    args = (t(d), t(p), t(np.float32(V)), t(F), t(s), 1, 1, 0, np.float32(0.3))
    keep_tail, st_tail = sd3.c_non_max_suppression_inds(*args, return_stats=True)
    st_tail = st_tail.copy()
    with N.option("nms3d_tail_batch", 0):
        keep_rounds, st_rounds = sd3.c_non_max_suppression_inds(*args, return_stats=True)
    refmods.stardist3d(); refmods.set_threads(1)
    ref_keep = refmods.stardist3d().c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(0.3))
    assert np.array_equal(keep_tail.cpu().numpy(), keep_rounds.cpu().numpy()) and np.array_equal(keep_rounds.cpu().numpy(), ref_keep)
    assert st_tail[4] <= st_rounds[4], (st_tail[4], st_rounds[4])       # never more host-driven rounds


def test_nms3d_neighbour_list_forms_agree(refmods):
    """single-pass neighbour lists (slots sized from the cell table, the default) vs count / scan / fill: the reference's survivors, the
    same number of list entries, with and without the tail batch and in the all-pairs configuration"""
    import torch
    from oracle import synth
    from stardist_amd.lib import _native as N, stardist3d as sd3
    rays = _rays(96)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    d, p, s, nobj = synth.s3d_nuclei(96, rays.vertices)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    refmods.stardist3d(); refmods.set_threads(1)
    for kd in (1, 0):
        n = len(d) if kd else 1500
        args = (t(d[:n]), t(p[:n]), t(np.float32(V)), t(F), t(s[:n]), 1, kd, 0, np.float32(0.3))
        ref_keep = refmods.stardist3d().c_non_max_suppression_inds(d[:n], p[:n], V, F, s[:n], 1, kd, 0, np.float32(0.3))
        entries = []
        for form in (1, 0):
            for tail in (1, 0):
                with N.option("nms3d_neighbours_single_pass", form), N.option("nms3d_tail_batch", tail):
                    keep, st = sd3.c_non_max_suppression_inds(*args, return_stats=True)
                assert np.array_equal(keep.cpu().numpy(), ref_keep), (kd, form, tail)
                entries.append(int(st[5]))
        assert len(set(entries)) == 1, entries


def test_nms3d_exact_volumes_carried_into_the_tail_batch(refmods, capfd):
    """"nms3d_defer_exact": the pairs the bounds leave undecided in rounds >= r are queued and evaluated by the tail batch's one pass
    (k_defer3 / k_seed3, pending candidates stay undecided): the reference's survivors for every r -- r = 1 included, the strongest
    form (every round defers; the rounds end early once everything left waits for a pending candidate) -- on the noisy random sets of
    the reference's own test (many pairs near the threshold) and on the nuclei set; the trace tells how many pairs were carried"""
    import re
    import torch
    from oracle import synth
    from stardist_amd.lib import _native as N, stardist3d as sd3
    rays = _rays(96)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    refmods.stardist3d(); refmods.set_threads(1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    sets = []
    for thr in (0.3, 0.6):
        d, p, s = _random_candidates((22, 33, 44), 96, 0.3, seed=96)
        sets.append(("random %.1f" % thr, d, p, s, thr, _ref_keep_random(refmods, (22, 33, 44), 96, 0.3, thr)))
    d, p, s, nobj = synth.s3d_nuclei(96, rays.vertices)
    sets.append(("nuclei", d, p, s, 0.3, refmods.stardist3d().c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(0.3))))
    report = {}
    for name, d, p, s, thr, ref_keep in sets:
        args = (t(d), t(p), t(np.float32(V)), t(F), t(s), 1, 1, 0, np.float32(thr))
        for r in (0, 1, 2, 3):
            capfd.readouterr()
            with N.option("nms3d_defer_exact", r), N.option("trace", 1):
                keep = sd3.c_non_max_suppression_inds(*args)
                torch.cuda.synchronize()
            out = capfd.readouterr().out
            assert np.array_equal(keep.cpu().numpy(), ref_keep), (name, r, int(keep.sum()), int(ref_keep.sum()))
            m = re.search(r"(\d+) pairs carried over", out)
            report[(name, r)] = int(m.group(1)) if m else None
    print("pairs carried into the tail batch:", report)
    assert all(report[(name, 0)] in (0, None) for name, *_ in sets), report          # None: the rounds ended without a tail batch
    assert any(report[(name, 1)] for name, *_ in sets), report                       # the mechanism ran


def test_nms3d_bounds_reuse_does_not_change_decisions(refmods):
    """"nms3d_bounds_reuse": the refined bounds pass takes the ray directions' boundary points from the coarse pass / casts them again:
    same survivors, same number of pairs decided by each bound and integrated exactly (the bounds are bit-identical)"""
    import torch
    from oracle import synth
    from stardist_amd.lib import _native as N, stardist3d as sd3
    rays = _rays(96)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    d, p, s, nobj = synth.s3d_nuclei(96, rays.vertices)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    args = (t(d), t(p), t(np.float32(V)), t(F), t(s), 1, 1, 0, np.float32(0.3))
    refmods.stardist3d(); refmods.set_threads(1)
    ref_keep = refmods.stardist3d().c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(0.3))
    res = {}
    for reuse, lean in ((1, 1), (0, 1), (1, 0), (0, 0)):
        # "nms3d_bounds_lean": the bounds-only launches without the seed / pos / orig tables behind the workspace (seven waves per CU) / with them
        with N.option("nms3d_bounds_reuse", reuse), N.option("nms3d_bounds_lean", lean):
            keep, st = sd3.c_non_max_suppression_inds(*args, return_stats=True)
        assert np.array_equal(keep.cpu().numpy(), ref_keep), (reuse, lean)
        res[reuse, lean] = [int(st[k]) for k in (0, 1, 2, 3, 6, 7, 11, 12, 13)]
    assert len(set(map(tuple, res.values()))) == 1, res


def test_nms3d_split_exact_does_not_change_survivors_or_volumes(refmods):
    """exact volumes by four waves per pair in a second pass (k_stage3x / k_stage4x) vs by the wave that evaluated the bounds: the
    same survivors as the reference, with and without the bound shortcuts, and bit-identical pair volumes"""
    import torch
    from oracle import synth
    from stardist_amd.lib import _native as N, stardist3d as sd3
    rays = _rays(96)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    d, p, s, nobj = synth.s3d_nuclei(96, rays.vertices)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    args = (t(d), t(p), t(np.float32(V)), t(F), t(s), 1, 1, 0, np.float32(0.3))
    refmods.stardist3d(); refmods.set_threads(1)
    ref_keep = refmods.stardist3d().c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(0.3))
    keeps = {}
    for split in (2, 3, 1, 0):          # 3 (default): second pass for every launch + small-footprint bounds pass; 2: stage 4 up to 16 384 pairs
        for bounds in (1, 0):
            with N.option("nms3d_split_exact", split), N.option("nms3d_volume_bounds", bounds):
                keeps[split, bounds] = sd3.c_non_max_suppression_inds(*args).cpu().numpy()
    for k, v in keeps.items():
        assert np.array_equal(v, ref_keep), k
    rs = np.random.RandomState(0)
    o = np.argsort(-s, kind="stable")[:4000]
    pairs = np.stack([rs.randint(0, 4000, 3000), rs.randint(0, 4000, 3000)], 1).astype(np.int32)
    pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    dd, pp = np.ascontiguousarray(d[o]), np.ascontiguousarray(p[o])
    close = np.linalg.norm(pp[pairs[:, 0]] - pp[pairs[:, 1]], axis=1) < 14
    pairs = np.ascontiguousarray(pairs[close][:600])
    vols = {}
    for split in (1, 0, 2):
        with N.option("nms3d_split_exact", split):
            vols[split] = [np.asarray(v) for v in sd3.hiv_pair_volumes(dd, pp, np.float32(V), F, pairs)]
    assert np.array_equal(vols[2][0], vols[0][0]) and np.array_equal(vols[2][1], vols[0][1])
    assert len(pairs) > 50 and (vols[1][0] > 0).sum() > 10
    assert np.array_equal(vols[1][0], vols[0][0]) and np.array_equal(vols[1][1], vols[0][1])


def test_nms3d_flags_and_edges(refmods):
    from stardist_amd.lib import stardist3d as sd3
    rays = _rays(32)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    d, p, s = _random_candidates((20, 30, 28), 32, 0.2, seed=5, prob_thresh=0.85)
    for bb, kd in [(1, 1), (0, 1), (1, 0)]:
        ref_keep = refmods.stardist3d().c_non_max_suppression_inds(d, p, V, F, s, bb, kd, 0, np.float32(0.3))
        keep = sd3.c_non_max_suppression_inds(d, p, V, F, s, bb, kd, 0, np.float32(0.3))
        assert np.array_equal(keep, ref_keep), (bb, kd)
    assert sd3.c_non_max_suppression_inds(d[:0], p[:0], V, F, s[:0], 1, 1, 0, 0.3).shape == (0,)
    assert sd3.c_non_max_suppression_inds(d[:1], p[:1], V, F, s[:1], 1, 1, 0, 0.3).tolist() == [True]


@pytest.mark.parametrize("noise", (0, .2, .6, .9))
@pytest.mark.parametrize("n_rays", (32, 65, 100))
def test_nms3d_accuracy_like_reference(refmods, noise, n_rays):
    """the reference's own test (tests/test_nms3D.py:60-83): NMS pinned against the rasteriser's IoU"""
    from stardist_amd.lib import stardist3d as sd3
    rays = _rays(n_rays)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    # NB: the expression yields a Fortran-ordered array; the reference natives read raw C-order memory
    # (its Python wrapper always passes np.ascontiguousarray, nms.py:365-366)
    dist = np.ascontiguousarray((10 * (1 + noise * np.sin(2 * np.pi * rays.vertices[:, :2].T))).astype(np.float32))
    points = np.array([(20, 20, 20), (20, 20, 23)], np.float32)
    shape = (40, 55, 66)
    one = np.ones(1, np.int32)
    m1 = sd3.c_polyhedron_to_label(dist[:1], points[:1], V, F, one, 0, 0, 0, 0, shape)
    m2 = sd3.c_polyhedron_to_label(dist[1:], points[1:], V, F, one, 0, 0, 0, 0, shape)
    iou = np.count_nonzero(m1 * m2) / min(np.count_nonzero(m1), np.count_nonzero(m2) + 1e-10)
    prob = np.array([1, .5], np.float32)
    k1 = sd3.c_non_max_suppression_inds(dist, points, V, F, prob, 1, 1, 0, np.float32(0.95 * iou))
    k2 = sd3.c_non_max_suppression_inds(dist, points, V, F, prob, 1, 1, 0, np.float32(1.05 * iou))
    assert k1.sum() == 1 and k2.sum() == 2
    r1 = refmods.stardist3d().c_non_max_suppression_inds(dist, points, V, F, prob, 1, 1, 0, np.float32(0.95 * iou))
    r2 = refmods.stardist3d().c_non_max_suppression_inds(dist, points, V, F, prob, 1, 1, 0, np.float32(1.05 * iou))
    assert np.array_equal(k1, r1) and np.array_equal(k2, r2)


def test_dist_to_volume_and_centroid_vs_reference(refmods):
    """analysis helpers of the native module (stardist3d.cpp:148-243), float tolerance 1e-5"""
    from stardist_amd.geometry.geom3d import dist_to_centroid, dist_to_volume
    rays = _rays(48)
    V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
    rng = np.random.RandomState(1)
    dist = (4 + 4 * rng.rand(9, 10, 11, 48)).astype(np.float32)
    m3 = refmods.stardist3d()
    assert np.allclose(dist_to_volume(dist, rays), m3.c_dist_to_volume(dist, V, F), rtol=1e-5, atol=1e-4)
    for mode in ("absolute", "relative"):
        assert np.allclose(dist_to_centroid(dist, rays, mode), m3.c_dist_to_centroid(dist, V, F, int(mode == "absolute")), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("n_rays,noise", [(96, 0.05), (96, 0.3), (32, 0.2)])
def test_pair_volumes_match_qhull(refmods, n_rays, noise):
    """stages 3 and 4 at the pair level (the analogue of the 2D Clipper probe): the wave-cooperative fp64 half-space-intersection
    volume vs the reference's qhull_overlap_kernel / qhull_overlap_convex_hulls (stardist3d_impl.cpp:830-939, float) on 12 000
    random overlapping pairs, incl. Qhull's error values (0 for an infeasible interior point of the kernels, 1e10 for the hulls)"""
    from stardist_amd.lib import stardist3d as sd3
    rays = _rays(n_rays)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    rng = np.random.RandomState(n_rays)
    n = 3000
    d = (8.0 * (1 + noise * rng.uniform(-1, 1, (n, n_rays)))).astype(np.float32)
    p = rng.uniform(20, 44, (n, 3)).astype(np.float32)
    i = rng.randint(0, n, 40000); j = rng.randint(0, n, 40000)
    sep = np.sqrt(((p[i] - p[j]) ** 2).sum(1))
    sel = np.flatnonzero((i != j) & (sep < 14))[:12000]
    pairs = np.stack([i[sel], j[sel]], 1).astype(np.int32)
    assert len(pairs) >= 10000
    rk, rh = refmods.pair_volumes(d, p, V, F, pairs)
    gk, gh = sd3.hiv_pair_volumes(d, p, V, F, pairs)
    # kernels: same zero pattern (Qhull error / empty) and float-level agreement elsewhere
    assert np.array_equal(rk == 0, gk.astype(np.float32) == 0), np.flatnonzero((rk == 0) != (gk.astype(np.float32) == 0))[:10]
    nz = rk != 0
    relk = np.abs(gk[nz] - rk[nz]) / np.maximum(np.abs(rk[nz]), 1e-3)
    bigh = rh > 1e9
    assert np.array_equal(bigh, gh > 1e9), np.flatnonzero(bigh != (gh > 1e9))[:10]
    relh = np.abs(gh[~bigh] - rh[~bigh]) / np.maximum(np.abs(rh[~bigh]), 1e-3)
    print("kernel volumes: %d non-zero, max rel diff %.3g; hull volumes: %d finite, max rel diff %.3g" % (nz.sum(), relk.max() if nz.any() else 0, (~bigh).sum(), relh.max()))
    assert nz.sum() > 1000 and (~bigh).sum() > 5000
    assert (relk.max() if nz.any() else 0) < 2e-6 and relh.max() < 2e-6         # the reference returns float32


@pytest.mark.parametrize("n_rays,aniso", [(96, None), (32, None), (64, (2.0, 1.0, 1.0)), (11, None), (300, None)])
def test_inside_polyhedron_cone_map_equals_full_loop(n_rays, aniso):
    """stage 5 of the NMS tests a voxel only against the faces whose cone can contain its direction (csrc/geom3d.h): on lattice
    points, random points, points on the faces / on the cone boundaries (vertices, edge planes) and next to the centre the result
    must equal the loop over every face (inside_polyhedron, stardist3d_impl.cpp:153-191), also when the map's preconditions fail
    (dist < 1, far-away coordinates) and it has to step aside"""
    import ctypes
    import torch
    from stardist_amd.lib import _native as N
    dev = torch.device("cuda:0")
    rays = _rays(n_rays, aniso)
    V = np.ascontiguousarray(rays.vertices, np.float32); Fc = np.ascontiguousarray(rays.faces, np.int32)
    tV, tF = torch.from_numpy(V).to(dev), torch.from_numpy(Fc).to(dev)
    rng = np.random.RandomState(n_rays)
    total = 0
    for case in range(8):
        radius = [9.0, 9.0, 3.0, 25.0, 0.7, 9.0, 9.0, 1.5][case]
        noise = [0.0, 0.3, 0.6, 0.1, 0.2, 0.9, 0.05, 0.5][case]
        c = rng.uniform(20, 40, 3).astype(np.float32)
        if case == 3:
            c = np.float32([9000.25, 31.5, 17.75])                        # beyond the coordinate bound: full loop
        if case == 6:
            c = np.float32([30, 31, 32])                                   # lattice centre: voxels exactly on cone boundaries
        d = (radius * (1 + noise * rng.uniform(-1, 1, n_rays))).astype(np.float32)
        d = np.maximum(d, 1e-3).astype(np.float32)
        ext = float(d.max() * np.abs(V).max()) + 2
        g = np.arange(-int(ext) - 1, int(ext) + 2, dtype=np.float32)
        lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) + np.round(c)
        rnd = (c + rng.uniform(-ext, ext, (20000, 3))).astype(np.float32)
        verts = (c + d[:, None] * V).astype(np.float32)
        tri = verts[Fc]                                                      # (F, 3, 3)
        w = rng.dirichlet((1, 1, 1), (len(Fc), 8)).astype(np.float32)       # points on the faces
        onface = np.einsum("fkt,ftd->fkd", w, tri).reshape(-1, 3)
        t = rng.uniform(0, 1.2, (n_rays, 8, 1)).astype(np.float32)           # along the rays (cone apex lines)
        onray = (c + t * d[:, None, None] * V[:, None, :]).reshape(-1, 3)
        e = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]])  # points in the side planes (centre, edge)
        u = rng.uniform(0, 1, (len(e), 4, 1)).astype(np.float32); sc = rng.uniform(0, 1.1, (len(e), 4, 1)).astype(np.float32)
        onside = (c + sc * ((u * e[:, None, 0] + (1 - u) * e[:, None, 1]) - c)).reshape(-1, 3)
        near = (c + rng.uniform(-0.6, 0.6, (2000, 3))).astype(np.float32)
        pts = np.ascontiguousarray(np.concatenate([lat, rnd, onface, onray, onside, near, c[None]]).astype(np.float32))
        tp = torch.from_numpy(pts).to(dev); td = torch.from_numpy(d).to(dev); tc = torch.from_numpy(c).to(dev)
        outs = []
        for use_map in (0, 1):
            o = torch.empty(len(pts), dtype=torch.uint8, device=dev)
            N.dcall(tp, "sd_inside_polyhedron_device", N.tptr(td), N.tptr(tc), n_rays, len(Fc), N.tptr(tV), N.tptr(tF), N.tptr(tp), len(pts), use_map,
                    N.tptr(o))
            outs.append(o.cpu().numpy())
        assert np.array_equal(outs[0], outs[1]), (case, int((outs[0] != outs[1]).sum()))
        assert 0 < outs[0].sum() < len(pts)
        total += len(pts)
    assert total > 100000


@pytest.mark.parametrize("n_rays,noise,thr", [(96, 0.3, 0.3), (32, 0.5, 0.5), (96, 0.3, 0.6)])
def test_nms3d_cone_map_does_not_change_survivors(refmods, monkeypatch, n_rays, noise, thr):
    """stage 5 with the cone map vs the loop over every face (option nms3d_cone_map = 0) vs the reference: same survivors and the same
    cascade counters (pairs rendered, suppressed by the rendered overlap) on candidate sets that do reach stage 5"""
    import torch
    from stardist_amd.lib import stardist3d as sd3
    rays = _rays(n_rays)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    d, p, s = _random_candidates((22, 33, 44), n_rays, noise, seed=n_rays)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    args = (t(d), t(p), t(np.float32(V)), t(F), t(s), 1, 1, 0, np.float32(thr))
    keep_map, st_map = sd3.c_non_max_suppression_inds(*args, return_stats=True)
    st_map = st_map.copy()
    from stardist_amd.lib import _native as N
    with N.option("nms3d_cone_map", 0):
        keep_full, st_full = sd3.c_non_max_suppression_inds(*args, return_stats=True)
    ref_keep = _ref_keep_random(refmods, (22, 33, 44), n_rays, noise, thr)
    assert np.array_equal(keep_map.cpu().numpy(), keep_full.cpu().numpy()) and np.array_equal(keep_full.cpu().numpy(), ref_keep)
    assert st_map[3] > 0, "no pair reached the render stage: %s" % st_map.tolist()
    assert np.array_equal(st_map[[0, 1, 2, 3, 6, 7]], st_full[[0, 1, 2, 3, 6, 7]])
