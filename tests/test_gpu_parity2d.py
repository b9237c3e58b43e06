"""GPU parity tests (2D): the HIP path through the C ABI vs the compiled reference (oracle/_ref)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _star_polys(rng, n, R, radius, noise, spread):
    """integer star polygons built with the reference's vertex arithmetic (stardist2d.cpp:447-471)"""
    ang = np.float32(2 * np.pi / R)
    k = np.arange(R, dtype=np.int32)
    s = np.sin((ang * k).astype(np.float32)).astype(np.float32)
    c = np.cos((ang * k).astype(np.float32)).astype(np.float32)
    d = (radius * (1 + noise * rng.uniform(-1, 1, (n, R)))).astype(np.float32)
    d = np.maximum(d, np.float32(1e-3))
    p = np.floor(rng.uniform(50, 50 + spread, (n, 2))).astype(np.float32)
    y = (p[:, :1] + d * s).astype(np.float32)
    x = (p[:, 1:] + d * c).astype(np.float32)
    return x.astype(np.int64).astype(np.int32), y.astype(np.int64).astype(np.int32)


@pytest.mark.parametrize("R,radius,noise", [(32, 10, 0.1), (32, 10, 0.9), (32, 3, 0.5), (11, 10, 0.3), (64, 20, 0.3), (100, 30, 0.6)])
def test_pair_area_matches_clipper(refmods, R, radius, noise):
    from stardist_amd.lib import stardist2d as sd2
    rng = np.random.RandomState(R * 1000 + int(radius))
    n = 4000
    xa, ya = _star_polys(rng, n, R, radius, noise, 12)
    xb, yb = _star_polys(rng, n, R, radius * 0.8, noise, 12)
    twice, flags = sd2.clip_pairs(xa, ya, xb, yb)
    assert not np.any(flags & 0xFF), "capacity overflow flags set"
    ref_area = np.array([refmods.clipper_area(xa[i], ya[i], xb[i], yb[i]) for i in range(n)], np.float32)
    mine = (0.5 * twice.astype(np.float32)).astype(np.float32)
    bad = np.flatnonzero(mine != ref_area)
    joins = np.flatnonzero(flags & 256)
    # every mismatch must be a pair that needed the join path
    assert set(bad.tolist()) <= set(joins.tolist()), (bad[:10], mine[bad[:10]], ref_area[bad[:10]])
    assert len(bad) == 0, "join-path mismatches: %d of %d (join pairs %d)" % (len(bad), n, len(joins))


@pytest.mark.parametrize("shape,R,thr", [((256, 256), 32, 0.4), ((512, 512), 32, 0.4), ((356, 299), 11, 0.5), ((114, 217), 32, 0.3)])
def test_nms2d_survivors_bit_exact(refmods, shape, R, thr):
    from oracle import synth
    from stardist_amd.lib import stardist2d as sd2
    d, p, s = synth.s2d_uniform(shape[0], shape[1], n_rays=R)
    ref_keep = refmods.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr))
    keep, stats = sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr), return_stats=True)
    assert keep.dtype == bool and keep.shape == ref_keep.shape
    diff = np.flatnonzero(keep != ref_keep)
    assert len(diff) == 0, "survivor mismatch at %s (pairs=%d joins=%d)" % (diff[:10], stats[0], stats[1])


@pytest.mark.parametrize("shape,R,radius,noise,thr", [((120, 110), 64, 12, 0.2, 0.4), ((160, 150), 96, 15, 0.5, 0.3), ((150, 140), 128, 20, 0.3, 0.5),
                                                       ((120, 110), 200, 25, 0.1, 0.4), ((90, 80), 48, 8, 0.1, 0.6), ((100, 100), 256, 18, 0.05, 0.4)])
def test_nms2d_survivors_bit_exact_many_rays(refmods, shape, R, radius, noise, thr):
    """more than 32 rays (round 6: until then only the pair-level probe had met them): no decision shortcut, no offset-ordered pair list --
    every pair goes to the bound-slot sweep of the next vertex capacity (64 / 128 / 256) and, with joins or beyond its capacities, to the
    general path of that capacity.  Same survivors as the compiled reference."""
    from oracle import synth
    from stardist_amd.lib import stardist2d as sd2
    d, p, s = synth.s2d_uniform(shape[0], shape[1], n_rays=R, radius=radius, noise=noise, seed=R)
    assert len(d) > 500
    ref_keep = refmods.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr))
    keep, stats = sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr), return_stats=True)
    diff = np.flatnonzero(keep != ref_keep)
    assert len(diff) == 0, "survivor mismatch at %s of %d (pairs=%d general path=%d)" % (diff[:10], len(d), stats[0], stats[1])


@pytest.mark.parametrize("shape,R,radius,noise", [((80, 80), 256, 18, 0.9), ((120, 110), 200, 25, 0.6)])
def test_nms2d_many_rays_beyond_the_capacities_fail_loudly_and_fast(shape, R, radius, noise):
    """polygons of 200 / 256 rays whose neighbouring rays differ by up to +-90 % exceed the fixed capacities of the general path (hundreds of
    local minima): the call has to say so -- not fault (round 6: 2048 workgroups of the 256-vertex kernel, 4 MB of scratch per wave, ended in a
    memory aperture violation) and not grind through every other overflowing pair for minutes (the launch stops at the first)"""
    import time
    from oracle import synth
    from stardist_amd.lib import _native, stardist2d as sd2
    d, p, s = synth.s2d_uniform(shape[0], shape[1], n_rays=R, radius=radius, noise=noise, seed=R)
    t0 = time.time()
    with pytest.raises(_native.NativeError, match="exceeded the general path's fixed capacities"):
        sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4))
    assert time.time() - t0 < 30
    # the library is usable afterwards
    d2, p2, s2 = synth.s2d_uniform(64, 64, n_rays=32)
    assert sd2.c_non_max_suppression_inds(d2, p2, 1, 1, 0, np.float32(0.4)).sum() > 0


class _ref_threads(object):
    """OpenMP threads of the compiled reference for one call; afterwards 16 at most (the GPU boxes have 256 hardware threads, which
    oversubscribe the reference's fine-grained OpenMP loops: tests that time out on that set their own count)"""

    def __init__(self, refmods, n): self.r, self.n = refmods, n

    def __enter__(self): self.r.set_threads(self.n)

    def __exit__(self, *a):
        import os
        self.r.set_threads(min(os.cpu_count() or 1, 16))


def _cands2d(rng, n, R, radius, noise, extent):
    d = (radius * (1 + noise * rng.uniform(-1, 1, (n, R)))).astype(np.float32)
    p = rng.uniform(0, extent, (n, 2)).astype(np.float32).round()
    s = rng.uniform(0, 1, n).astype(np.float32)
    o = np.argsort(s, kind="stable")[::-1]
    return np.ascontiguousarray(d[o]), np.ascontiguousarray(p[o]), np.ascontiguousarray(s[o])


@pytest.mark.parametrize("R,radius,noise,extent,n,thr", [(32, 0.4, 0.5, 12, 600, 0.3), (32, 1.1, 0.8, 20, 800, 0.5), (8, 0.8, 0.3, 10, 400, 0.1),
                                                          (32, 400, 0.2, 3000, 300, 0.4), (32, 1500, 0.1, 9000, 200, 0.3), (16, 6000, 0.3, 30000, 150, 0.5),
                                                          (32, 40, 0.3, 60000, 700, 0.4)])
def test_nms2d_extreme_sizes(refmods, R, radius, noise, extent, n, thr):
    """polygons far below one pixel (every vertex truncates to the centre or its neighbours: zero areas, the reference divides by
    area + 1e-10) and far beyond the windows of the fast paths (16-bit relative coordinates of the first sweep tier: +-32 767; exact
    float predicates of the decision shortcut: +-1 023), and ordinary polygons at coordinates up to 60 000: same survivors"""
    from stardist_amd.lib import stardist2d as sd2
    rng = np.random.RandomState(int(radius * 10) + R)
    d, p, s = _cands2d(rng, n, R, radius, noise, extent)
    with _ref_threads(refmods, 2):
        ref_keep = refmods.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr))
    for strict in (0, 1):
        from stardist_amd.lib import _native
        with _native.option("nms2d_strict", strict):
            keep = sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr))
        diff = np.flatnonzero(keep != ref_keep)
        assert len(diff) == 0, (R, radius, strict, diff[:10], int(ref_keep.sum()), int(keep.sum()))
    assert 0 < ref_keep.sum() <= n


@pytest.mark.parametrize("thr", [0.0, -0.1, 1.0, 1.5, 1e-6, 0.999999])
def test_nms2d_threshold_edges(refmods, thr):
    """thresholds at and beyond the ends of [0, 1] (stardist2d.cpp:580-581 `overlap > thr`; a negative one suppresses on any candidate pair
    the bounding boxes let through, one above 1 never)"""
    from oracle import synth
    from stardist_amd.lib import stardist2d as sd2
    d, p, s = synth.s2d_uniform(64, 56, n_rays=32, radius=7, noise=0.3, seed=11)
    with _ref_threads(refmods, 2):               # (the reference's OpenMP loop over a few hundred candidates: more threads only spin)
        ref_keep = refmods.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr))
    keep = sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr))
    assert np.array_equal(keep, ref_keep), (thr, int(keep.sum()), int(ref_keep.sum()), np.flatnonzero(keep != ref_keep)[:10])


@pytest.mark.parametrize("flags", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_nms2d_flags(refmods, flags):
    from oracle import synth
    from stardist_amd.lib import stardist2d as sd2
    d, p, s = synth.s2d_uniform(128, 160, n_rays=32, prob_thresh=0.8)
    kd, bb = flags
    for thr in (0.0, 0.3, 0.7):
        ref_keep = refmods.stardist2d().c_non_max_suppression_inds(d, p, kd, bb, 0, np.float32(thr))
        keep = sd2.c_non_max_suppression_inds(d, p, kd, bb, 0, np.float32(thr))
        assert np.array_equal(keep, ref_keep), (flags, thr)


def test_nms2d_edge_cases(refmods):
    from stardist_amd.lib import stardist2d as sd2
    assert sd2.c_non_max_suppression_inds(np.zeros((0, 32), np.float32), np.zeros((0, 2), np.float32), 1, 1, 0, 0.4).shape == (0,)
    d = np.full((1, 32), 5, np.float32); p = np.full((1, 2), 20, np.float32)
    assert sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, 0.4).tolist() == [True]
    # identical polygons: the second is suppressed
    d = np.full((2, 32), 5, np.float32); p = np.full((2, 2), 20, np.float32)
    assert sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, 0.4).tolist() == refmods.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4)).tolist()
    # degenerate tiny distances (clamped 1e-3 as base.py:556 does)
    d = np.full((50, 32), 1e-3, np.float32); p = np.random.RandomState(0).randint(5, 20, (50, 2)).astype(np.float32)
    assert np.array_equal(sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, 0.4), refmods.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4)))


@pytest.mark.parametrize("n_rays,grid", [(32, (1, 1)), (17, (2, 2)), (64, (1, 4)), (4, (3, 1))])
def test_star_dist2d_bit_exact(refmods, n_rays, grid):
    from oracle import synth
    from stardist_amd.lib import stardist2d as sd2
    lbl, _, _ = synth.s2d_nuclei_labels(200, 231, seed=3)
    ref_d = refmods.stardist2d().c_star_dist(lbl, n_rays, grid[0], grid[1])
    d = sd2.c_star_dist(lbl, n_rays, grid[0], grid[1])
    assert d.shape == ref_d.shape and d.dtype == np.float32
    assert np.array_equal(d, ref_d), np.abs(d - ref_d).max()


def test_raster2d_matches_port(refmods):
    from oracle import port, synth
    from stardist_amd.lib import stardist2d as sd2
    d, p, s = synth.s2d_uniform(96, 128, n_rays=32, prob_thresh=0.97)
    coord = port.dist_to_coord(d, p)
    ind = np.argsort(s, kind="stable")
    ref_lbl = port.polygons_to_label_coord(coord[ind], (96, 128), labels=ind)
    lbl = sd2.c_polygons_to_label(coord[ind], ind.astype(np.int32), (96, 128))
    assert lbl.dtype == np.int32
    assert np.array_equal(lbl, ref_lbl), np.count_nonzero(lbl != ref_lbl)


def test_reference_style_c_abi_2d(refmods):
    """_LIB_non_maximum_suppression_2d / _LIB_star_dist / _LIB_polygon_to_label (plain C ABI, host pointers) == the sd_* entry points"""
    import ctypes
    from oracle import port, synth
    from stardist_amd.lib import _native as N, stardist2d as sd2
    d, p, s = synth.s2d_uniform(96, 128)
    keep = sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4))
    res = np.zeros(len(d), np.bool_)
    N.lib()._LIB_non_maximum_suppression_2d(N.ptr(d), N.ptr(p), len(d), d.shape[1], ctypes.c_float(0.4), 1, 1, 0, res.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(res, keep)
    lbl = synth.s2d_nuclei_labels(64, 80, seed=2)[0].astype(np.uint16)
    dst = np.zeros((64, 80, 16), np.float32)
    N.lib()._LIB_star_dist(N.ptr(lbl), 64, 80, 16, 1, 1, N.ptr(dst))
    assert np.array_equal(dst, refmods.stardist2d().c_star_dist(lbl, 16, 1, 1))
    coord = np.ascontiguousarray(port.dist_to_coord(d[keep], p[keep]), np.float32)
    ids = np.arange(len(coord), dtype=np.int32)
    out = np.zeros((96, 128), np.int32)
    N.lib()._LIB_polygon_to_label(N.ptr(coord), N.ptr(ids), len(coord), coord.shape[2], 96, 128, N.ptr(out))
    assert np.array_equal(out, sd2.c_polygons_to_label(coord, ids, (96, 128)))


@pytest.mark.parametrize("name", ("stars32", "stars8_small", "stars64_big", "stars5_int"))
def test_raster2d_equals_reference_with_real_skimage(name):
    """HIP rasteriser against golden label images made by the reference's own geom2d code on the real scikit-image
    (tests/golden/make_raster2d_golden.py): bit-identical"""
    import os
    from stardist_amd.geometry.geom2d import polygons_to_label
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raster2d_reference.npz"))
    lbl = polygons_to_label(g[name + "_dist"], g[name + "_points"], tuple(g[name + "_shape"]), prob=g[name + "_prob"], thr=0.2)
    assert np.array_equal(np.asarray(lbl), g[name + "_labels"])


def test_raster2d_lattice_and_degenerate_cases_vs_skimage():
    import os
    from stardist_amd.lib import stardist2d as sd2
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raster2d_reference.npz"))
    coord, shape = np.ascontiguousarray(g["explicit_coord"], np.float32), tuple(int(v) for v in g["explicit_shape"])
    lbl = sd2.c_polygons_to_label(coord, np.arange(len(coord), dtype=np.int32), shape)
    assert np.array_equal(lbl, g["explicit_labels"])
    for i in range(len(coord)):
        one = sd2.c_polygons_to_label(coord[i:i + 1], np.zeros(1, np.int32), shape)
        assert np.array_equal(one > 0, g["explicit_mask%d" % i]), i


def test_nms2d_pair_kernel_forms_agree(refmods):
    """tier 1 of the pair kernel in its three launch forms -- 16-bit coordinates relative to the pair's origin with recomputed slopes
    (the default: six waves per CU), 32-bit coordinates with stored slopes (four waves), half-filled 32-lane waves -- returns the
    reference's survivors; large absolute coordinates (a candidate set far from the image origin) do not disturb the relative form"""
    from oracle import synth
    from stardist_amd.lib import _native as N, stardist2d as sd2
    d, p, s = synth.s2d_uniform(512, 512)
    p_far = (p + np.float32(12000)).astype(np.float32)
    refmods.set_threads(8)
    want = refmods.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4))
    want_far = refmods.stardist2d().c_non_max_suppression_inds(d, p_far, 1, 1, 0, np.float32(0.4))
    try:
        for lanes in (64, 6464, 32):
            N.check(N.lib().sd_set_option(b"nms2d_pair_lanes", lanes))
            assert np.array_equal(sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4)), want), lanes
            assert np.array_equal(sd2.c_non_max_suppression_inds(d, p_far, 1, 1, 0, np.float32(0.4)), want_far), lanes
    finally:
        N.check(N.lib().sd_set_option(b"nms2d_pair_lanes", 64))


# ---- the decision shortcut (csrc/area_bounds.h): enclosure of Clipper's area from regular arithmetic
@pytest.mark.parametrize("R,radius,noise,spread", [(32, 10, 0.1, 12), (32, 10, 0.3, 25), (32, 4, 0.3, 6), (16, 25, 0.2, 30), (32, 10, 0.9, 12), (7, 12, 0.2, 14)])
def test_area_enclosure_probe_matches_statement(R, radius, noise, spread):
    """the GPU probe against the numpy statement of the same arithmetic (tests/_area_exact.py): area, crossing count, usability"""
    from _area_exact import band as band_np, edge_stats, exact_area, near_pairs, near_strips, plain
    from stardist_amd.lib import stardist2d as sd2
    rng = np.random.RandomState(R * 7 + int(radius))
    n = 3000
    xa, ya = _star_polys(rng, n, R, radius, noise, spread)
    xb, yb = _star_polys(rng, n, R, radius * 0.8, noise, spread)
    area, band, usable, K, T = sd2.area_bounds_pairs(xa, ya, xb, yb)
    A, Ks, ok, _, _ = exact_area(*(v.astype(np.int64) for v in (xa, ya, xb, yb)))
    want = ok & plain(xa.astype(np.int64), ya.astype(np.int64)) & plain(xb.astype(np.int64), yb.astype(np.int64))
    assert np.array_equal(usable, want), (np.flatnonzero(usable != want)[:10], usable.mean(), want.mean())
    assert np.array_equal(K[usable], Ks[usable])
    assert np.array_equal(T[usable], near_pairs(*(v.astype(np.int64) for v in (xa, ya, xb, yb)))[usable])
    assert np.all(np.abs(area[usable] - A[usable]) <= 2e-3 + 1e-6 * A[usable]), np.abs(area[usable] - A[usable]).max()
    # the band itself: 0.5 K + max(NEAR_W T, STRIP_W S) per unit of (lmax_P + lmax_Q), + 0.75 + the float term of the pair's extent about
    # the centre of P's box (round 6: the strip term)
    i64 = [v.astype(np.int64) for v in (xa, ya, xb, yb)]
    la, pa = edge_stats(i64[0], i64[1]); lb, pb = edge_stats(i64[2], i64[3])
    ox = (i64[0].min(1) + i64[0].max(1)) >> 1; oy = (i64[1].min(1) + i64[1].max(1)) >> 1
    ext = np.maximum.reduce([np.abs(i64[0] - ox[:, None]).max(1), np.abs(i64[1] - oy[:, None]).max(1), np.abs(i64[2] - ox[:, None]).max(1),
                             np.abs(i64[3] - oy[:, None]).max(1)]).astype(np.float64)
    want_band = band_np(Ks, near_pairs(*i64), la, lb, ext, pa, pb, near_strips(*i64))
    assert np.allclose(band[usable], want_band[usable], rtol=2e-5, atol=1e-3), np.abs(band[usable] - want_band[usable]).max()


@pytest.mark.parametrize("R,radius,noise,spread,scale", [(32, 10, 0.1, 12, 0.8), (32, 10, 0.03, 6, 1.0), (32, 10, 0.03, 3, 0.97), (32, 20, 0.05, 6, 0.95),
                                                         (32, 10, 0.3, 25, 0.8), (32, 4, 0.3, 6, 0.8), (32, 2.5, 0.3, 4, 1.0), (16, 25, 0.2, 30, 0.8),
                                                         (32, 40, 0.1, 60, 0.9), (32, 10, 0.9, 12, 0.8), (24, 200, 0.2, 300, 0.8)])
def test_area_enclosure_contains_clipper_area(R, radius, noise, spread, scale):
    """for every pair the shortcut may use, the area of the Clipper-exact sweep lies inside the band (400 k pairs per family)"""
    from stardist_amd.lib import stardist2d as sd2
    rng = np.random.RandomState(R * 13 + int(radius * 10) + int(noise * 100))
    n = 400000
    xa, ya = _star_polys(rng, n, R, radius, noise, spread)
    xb, yb = _star_polys(rng, n, R, radius * scale, noise, spread)
    twice, flags = sd2.clip_pairs(xa, ya, xb, yb)
    assert not np.any(flags & 0xFF)
    area, band, usable, K, T = sd2.area_bounds_pairs(xa, ya, xb, yb)
    d = np.abs(0.5 * twice.astype(np.float64) - area.astype(np.float64))
    worst = (d[usable] / band[usable]).max() if usable.any() else 0.0
    print("R=%d radius=%g noise=%g spread=%g scale=%g: usable %.4f, crossings mean %.1f, near pairs mean %.1f, max |A_clipper - A| / band %.3f (max |d| %.2f)"
          % (R, radius, noise, spread, scale, usable.mean(), K[usable].mean() if usable.any() else 0, T[usable].mean() if usable.any() else 0, worst,
             d[usable].max() if usable.any() else 0))
    assert worst <= 0.5, worst                               # a bound with room to spare, not a fit


@pytest.mark.parametrize("thr", [0.3, 0.5])
def test_nms2d_area_bounds_on_off(refmods, thr):
    from oracle import synth
    from stardist_amd.lib import _native as N, stardist2d as sd2
    d, p, s = synth.s2d_uniform(384, 384, n_rays=32, prob_thresh=0.85)
    ref_keep = refmods.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr))
    for on in (1, 0):
        with N.option("nms2d_area_bounds", on):
            keep, stats = sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr), return_stats=True)
        assert np.array_equal(keep, ref_keep), (on, np.flatnonzero(keep != ref_keep)[:10])
        assert (stats[9] > 0) == bool(on), stats[:10]


@pytest.mark.parametrize("shape,R,thr", [((384, 384), 32, 0.4), ((300, 280), 32, 0.7), ((356, 299), 11, 0.5)])
def test_nms2d_defer_undecided_settings_agree(refmods, shape, R, thr):
    """the pairs the enclosure leaves undecided swept in their round (0), deferred to the tail batch from round 2 (default) or from round 1:
    the compiled reference's survivors every time"""
    from oracle import synth
    from stardist_amd.lib import _native as N, stardist2d as sd2
    d, p, s = synth.s2d_uniform(shape[0], shape[1], n_rays=R, prob_thresh=0.85)
    ref_keep = refmods.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr))
    deferred = {}
    for opt in (0, 2, 1):
        with N.option("nms2d_defer_undecided", opt):
            keep, stats = sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr), return_stats=True)
        assert np.array_equal(keep, ref_keep), (opt, np.flatnonzero(keep != ref_keep)[:10])
        deferred[opt] = int(stats[10])
    assert deferred[0] == 0


@pytest.mark.parametrize("grid", [(1, 1), (2, 2)])
@pytest.mark.parametrize("shape,R", [((356, 299), 11), ((114, 217), 32)])
@pytest.mark.parametrize("max_bbox_search", [1, 0])
def test_nms2d_old_equals_reference_old_and_new(refmods, shape, R, grid, max_bbox_search):
    """c_non_max_suppression_inds_old (stardist2d.cpp:173-386) through the C ABI: same keep flags as the compiled reference's _old on the
    seed-42 candidates of the reference's own old == new test (tests/test_nms2D.py:78-110), and -- that test's statement -- the same
    survivors as the new NMS (grid (1,1), where both see the same integer polygons)"""
    from oracle import port, synth
    from stardist_amd.lib import stardist2d as sd2
    m = refmods.stardist2d()
    refmods.set_threads(8)           # the reference's _old runs a collapse(2) dynamic OpenMP loop per polygon: 256 host threads oversubscribe it
    dist, prob = synth.s2d_uniform(shape[0], shape[1], n_rays=R, dense=True)
    dist, prob = dist[::grid[0], ::grid[1]], prob[::grid[0], ::grid[1]]
    mask = port.ind_prob_thresh(prob, 0.9, b=2)
    pts = np.stack(np.where(mask), 1)
    d, s = dist[mask], prob[mask]
    ind = np.argsort(s, kind="stable")[::-1]
    d, s, pts = d[ind], s[ind], pts[ind]
    coord = port.dist_to_coord(d, pts * np.array(grid))
    polys = np.ascontiguousarray(coord.astype(np.int32))
    if max_bbox_search:
        mapping = -np.ones(mask.shape, np.int32)
        mapping.flat[np.flatnonzero(mask)[ind]] = range(len(ind))
    else:
        mapping = np.empty((0, 0), np.int32)
    for thr in (0.3, 0.4):
        ref_old = m.c_non_max_suppression_inds_old(polys, mapping, np.float32(thr), np.int32(max_bbox_search), np.int32(grid[0]),
                                                   np.int32(grid[1]), np.int32(0))
        mine = sd2.c_non_max_suppression_inds_old(polys, mapping, np.float32(thr), np.int32(max_bbox_search), np.int32(grid[0]),
                                                  np.int32(grid[1]), np.int32(0))
        assert mine.dtype == bool and np.array_equal(mine, ref_old), np.flatnonzero(mine != ref_old)[:10]
        if grid == (1, 1):
            new = sd2.c_non_max_suppression_inds(np.ascontiguousarray(d), np.ascontiguousarray(pts.astype(np.float32)), 1, 1, 0, np.float32(thr))
            assert np.array_equal(new, mine)


def test_nms2d_old_python_path_equals_new(refmods):
    """tests/test_nms2D.py:78-110 replayed on the mirror's own functions: _dist_to_coord_old -> _non_maximum_suppression_old against
    non_maximum_suppression, points equal and foreground of the two label images equal"""
    from oracle import synth
    from stardist_amd.geometry.geom2d import _dist_to_coord_old, _polygons_to_label_old, polygons_to_label
    from stardist_amd.nms import _non_maximum_suppression_old, non_maximum_suppression
    for shape, R, grid in (((356, 299), 11, (1, 1)), ((114, 217), 32, (1, 1))):
        dist, prob = synth.s2d_uniform(shape[0], shape[1], n_rays=R, dense=True)
        prob, dist = prob[::grid[0], ::grid[1]], dist[::grid[0], ::grid[1]]
        coord = _dist_to_coord_old(dist, grid=grid)
        inds1 = _non_maximum_suppression_old(coord, prob, prob_thresh=0.9, grid=grid, nms_thresh=0.3)
        points1 = inds1 * np.array(grid)
        points1 = points1[np.argsort(prob[tuple(inds1.T)], kind="stable")[::-1]]
        points2, probi2, disti2 = non_maximum_suppression(dist, prob, grid=grid, prob_thresh=0.9, nms_thresh=0.3)
        img1 = _polygons_to_label_old(coord, prob, inds1, shape=shape)
        img2 = polygons_to_label(disti2, points2, shape=shape)
        assert len(points1) == len(points2)
        assert np.allclose(points1, points2)
        assert np.allclose(img1 > 0, img2 > 0)


def test_nms2d_neighbour_list_forms_agree(refmods):
    """the single-pass neighbour lists (slots sized from the cell table, option nms2d_neighbours_single_pass = 1, the default) and the two-pass
    count / scan / fill form give the reference's survivors -- incl. the all-pairs flags and a set far from the origin"""
    from oracle import synth
    from stardist_amd.lib import _native
    from stardist_amd.lib import stardist2d as sd2
    for shape, R, thr, off in (((300, 280), 32, 0.4, 0.0), ((200, 231), 17, 0.3, 9000.0)):
        d, p, s = synth.s2d_uniform(shape[0], shape[1], n_rays=R, prob_thresh=0.85)
        p = np.ascontiguousarray(p + np.float32(off))
        for kd, bb in ((1, 1), (0, 1), (1, 0)):
            ref_keep = refmods.stardist2d().c_non_max_suppression_inds(d, p, kd, bb, 0, np.float32(thr))
            for form in (1, 0):
                with _native.option("nms2d_neighbours_single_pass", form):
                    keep, st = sd2.c_non_max_suppression_inds(d, p, kd, bb, 0, np.float32(thr), return_stats=True)
                assert np.array_equal(keep, ref_keep), (shape, kd, bb, form, np.flatnonzero(keep != ref_keep)[:8])
