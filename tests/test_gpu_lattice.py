"""GPU: the natives on LATTICE-ALIGNED inputs -- integer centres, integer / half-integer ray lengths, few rays, many coincident and
one-pixel-shifted shapes (oracle/synth.py lattice_candidates_*): coincident edges and vertices for the Clipper-exact sweep and the area
band, voxels exactly on faces for the 3D predicates, polygon vertices on pixel centres for the rasterisers -- the tie cases random float
inputs never produce.  Goldens: the compiled reference, and the reference's Python rasteriser loop on the real scikit-image
(tests/golden/make_lattice_golden.py)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
G = dict(np.load(os.path.join(ROOT, "tests", "golden", "lattice_reference.npz")))
G.update(dict(np.load(os.path.join(ROOT, "tests", "golden", "lattice_reference_more.npz"))))      # round 6: symmetric ray sets (make_lattice_golden_more.py)
SHAPE2D, SIZE3D = (96, 96), 48
CASES2D = [(R, fam, 0) for R in (4, 8, 16, 32) for fam in ("const", "int", "half")]
# Rays_Cartesian: its pole rays differ by 1e-12 and COINCIDE in float32 (degenerate triangles at both poles; the reference's own Qhull calls
# print precision warnings on these polyhedra and its cascade runs on its error paths: DESIGN.md section 4 item 3a).  Known limit until round 5
# (xfail); followed since round 6: the hull of a point set with coincident / collinear points (k_hull's exhaustive search lost the facets
# whose three lowest-indexed points coincide), the rendered overlap over the WHOLE box of the first polyhedron (a zero-volume tetrahedron is
# "inside" on its whole plane).
RAYS3D = ("octo", "golden32", "golden32_aniso", "cartesian_8_5", "octo1", "octo2", "tetra3", "cartesian_11_5", "golden96")


def _rays_of(name):
    from make_lattice_golden import rays_of
    from make_lattice_golden_more import RAYS3D_MORE, rays_of_more
    return rays_of_more(name) if name in RAYS3D_MORE else rays_of(name)


@pytest.mark.parametrize("strict", [0, 1])
@pytest.mark.parametrize("R,fam,seed", CASES2D)
def test_nms2d_lattice_polygons(R, fam, seed, strict):
    """keep flags == compiled reference, with the default pair decisions (area band where a polygon is robustly simple) and with every
    pair swept (nms2d_strict)"""
    from oracle import synth
    from stardist_amd.lib import _native, stardist2d as sd2
    d, p, s = synth.lattice_candidates_2d(R, fam, seed, shape=SHAPE2D)
    for thr in (0.3, 0.5):
        want = np.unpackbits(G["nms2d_%d_%s_%d_%.1f" % (R, fam, seed, thr)])[:len(d)].astype(bool)
        with _native.option("nms2d_strict", strict):
            keep = sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr))
        diff = np.flatnonzero(keep != want)
        assert len(diff) == 0, (R, fam, thr, strict, len(diff), diff[:8])


@pytest.mark.parametrize("R,fam,seed", CASES2D)
def test_raster2d_lattice_polygons(R, fam, seed):
    """label image == the reference's loop over the real skimage.draw.polygon (vertices and edges through pixel centres)"""
    from oracle import synth
    from stardist_amd.geometry import polygons_to_label
    d, p, s = synth.lattice_candidates_2d(R, fam, seed, shape=SHAPE2D)
    keep = np.unpackbits(G["nms2d_%d_%s_%d_%.1f" % (R, fam, seed, 0.3)])[:len(d)].astype(bool)
    lab = np.asarray(polygons_to_label(d[keep], p[keep], SHAPE2D, prob=s[keep]))
    want = G["raster2d_%d_%s_%d" % (R, fam, seed)].astype(np.int32)
    assert np.array_equal(lab, want), (R, fam, int((lab != want).sum()))


def _decided_at_a_midpoint_on_a_hull_facet(j, want, p, d, V, tol=1e-9):
    """candidate j meets a better-scored survivor i (reference's flags) such that (c_i + c_j) / 2 lies on the boundary of the hull of i or of j"""
    from _hull import on_hull_boundary
    P = np.asarray(p, np.float32)
    for i in np.flatnonzero(want[:j]):
        if np.abs(P[i] - P[j]).max() > 2 * float(np.asarray(d).max()) + 2:
            continue
        mid = 0.5 * (P[i].astype(np.float64) + P[j])
        if on_hull_boundary(mid[None], P[[i, j]], np.asarray(d)[[i, j]], V, tol=tol).any():
            return True
    return False


@pytest.mark.parametrize("fam", ["const", "int"])
@pytest.mark.parametrize("name", RAYS3D)
def test_nms3d_lattice_polyhedra(name, fam):
    from oracle import synth
    from stardist_amd.lib import stardist3d as sd3
    rays = _rays_of(name)
    V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
    d, p, s = synth.lattice_candidates_3d(len(V), fam, size=SIZE3D)
    for thr in (0.2, 0.4):
        want = np.unpackbits(G["nms3d_%s_%s_%.1f" % (name, fam, thr)])[:len(d)].astype(bool)
        keep, st = sd3.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(thr), return_stats=True)
        diff = np.flatnonzero(keep != want)
        # One class of flags is exempt, of the same kind as the hull-facet voxels of the rasteriser: the hull
        # stage asks whether the MIDPOINT of the two centres is strictly inside both hulls (qh_sethalfspace, 1e10 if not); when that midpoint
        # lies EXACTLY on a hull facet, Qhull's answer is the sign of a 1e-16 residual of its normalised plane (whose bits depend on the order in
        # which quickhull added the points).  Found on the plain octahedron (octo1 / const / 0.4: one flag of the 10 000 of these sets).
        diff = np.array([j for j in diff if not _decided_at_a_midpoint_on_a_hull_facet(j, want, p, d, V)], int)
        assert len(diff) == 0, (name, fam, thr, len(diff), diff[:8], st.tolist())
        assert len(np.flatnonzero(keep != want)) <= 2


@pytest.mark.parametrize("name,fam,mode,mname", [(n, f, m, mn) for n in RAYS3D for f in ("const", "int") for m, mn in ((0, "full"), (1, "kernel"))])
def test_raster3d_lattice_polyhedra(name, fam, mode, mname):
    """voxel for voxel; mode "full" except voxels exactly on the hull of a polyhedron that covers them (tests/_hull.py: the reference's answer
    there is rounding noise of Qhull's planes; measured on these sets: 57 - 173 of 110 592 voxels, every one of them on a hull facet)"""
    from _hull import on_hull_boundary
    from oracle import synth
    from stardist_amd.lib import stardist3d as sd3
    rays = _rays_of(name)
    V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
    d, p, s = synth.lattice_candidates_3d(len(V), fam, size=SIZE3D)
    keep = np.unpackbits(G["nms3d_%s_%s_%.1f" % (name, fam, 0.2)])[:len(d)].astype(bool)
    lab = np.arange(1, keep.sum() + 1, dtype=np.int32)
    vol = np.asarray(sd3.c_polyhedron_to_label(d[keep], p[keep], V, F, lab, np.int32(mode), np.int32(0), np.int32(0), np.int32(0), (SIZE3D,) * 3)).astype(np.int32)
    want = G["raster3d_%s_%s_%s" % (name, fam, mname)].astype(np.int32)
    diff = np.argwhere(vol != want)
    print("raster3d %s %s %s: %d of %d voxels differ" % (name, fam, mname, len(diff), vol.size))
    if mode == 0 and len(diff):
        assert on_hull_boundary(diff, p[keep], d[keep], V).all(), (name, fam, len(diff), diff[:8])
        assert len(diff) <= 8 * int(keep.sum())
    else:
        assert len(diff) == 0, (name, fam, mname, len(diff), diff[:8])


@pytest.mark.parametrize("fam", ["const", "int"])
def test_cartesian_pair_volumes_match_qhull(refmods, fam):
    """Rays_Cartesian on the lattice sets, both Qhull stages pair by pair against the compiled reference (oracle shim): the kernel stage
    returns Qhull's error value 0 for EVERY pair (zero-normal half-spaces of the degenerate pole faces: qh_sethalfspace rejects them), the
    hull stage has the same error pattern (1e10: midpoint of the centres not inside both hulls) and, where finite, the same volume.
    Until round 5 the hulls of these point sets missed the facets at the poles (volumes up to 10x too large), and two polyhedra with
    identical distances whose centres differ along the pole axis only -- the vertical band of facets is then shared bit for bit -- had
    every shared facet counted twice, once or not at all (coincident half-spaces: DESIGN.md section 4 item 3a)."""
    from make_lattice_golden import rays_of
    from oracle import synth
    from stardist_amd.lib import stardist3d as sd3
    rays = rays_of("cartesian_8_5")
    V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
    d, p, s = synth.lattice_candidates_3d(len(V), fam, size=SIZE3D)
    P = p.astype(np.float32)
    ii, jj = np.triu_indices(len(d), 1)
    sep = np.sqrt(((P[ii] - P[jj]) ** 2).sum(1))
    sel = np.flatnonzero(sep < 12)[:2000]
    pairs = np.stack([ii[sel], jj[sel]], 1).astype(np.int32)
    rk, rh = refmods.pair_volumes(d, P, V, F, pairs)
    gk, gh = sd3.hiv_pair_volumes(d, P, V, F, pairs)
    assert (rk == 0).all() and (np.asarray(gk) == 0).all()
    assert np.array_equal(rh > 1e9, gh > 1e9), np.flatnonzero((rh > 1e9) != (gh > 1e9))[:10]
    symmetric = (P[pairs[:, 0], 1] == P[pairs[:, 1], 1]) & (P[pairs[:, 0], 2] == P[pairs[:, 1], 2]) & (d[pairs[:, 0]] == d[pairs[:, 1]]).all(1)
    fin = rh < 1e9
    rel = np.abs(gh[fin] - rh[fin]) / np.maximum(np.abs(rh[fin]), 1e-3)
    print("cartesian %s: %d pairs, %d with a finite hull volume (max rel diff %.3g), %d of them with shared facet planes" % (fam, len(pairs), int(fin.sum()), rel.max(), int((symmetric & fin).sum())))
    assert fin.sum() > 500 and rel.max() < 2e-6


@pytest.mark.parametrize("name", ["cartesian_8_5", "octo", "golden32"])
def test_volumes_of_shifted_copies_with_shared_facet_planes(refmods, name):
    """Two copies of ONE polyhedron (constant distances): a shift that lies in a facet plane (Rays_Cartesian along its pole axis, the
    octahedron along (1, 1, 0)) makes that plane a half-space of both, bit for bit -- and no shift at all every plane.  Each twin cuts the
    other's face with a trace line a = b = 0, e = +-1 ulp; until round 6 rounding decided whether such a face counted twice, once or not at
    all (285.8 for 321.7; coincident copies 223 for 386).  Both Qhull stages against the compiled reference."""
    from make_lattice_golden import rays_of
    from stardist_amd.lib import stardist3d as sd3
    rays = rays_of(name)
    V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
    shifts = [(0, 0, 0), (1e-5, 0, 0), (1, 0, 0), (1.00001, 0, 0), (1, 1e-5, 0), (2, 0, 0), (0, 1, 0), (1, 1, 0), (1, 1.00001, 0), (3, 0, 0), (0.5, 0, 0), (0, 2, 1), (1, 1, 1)]
    for dist in (5.0, 4.5):
        c0 = np.array([24, 24, 24], np.float32)
        P = np.array([c0] + [c0 + np.array(sh, np.float32) for sh in shifts], np.float32)
        d = np.full((len(P), len(V)), dist, np.float32)
        pairs = np.array([[0, k + 1] for k in range(len(shifts))], np.int32)
        rk, rh = refmods.pair_volumes(d, P, V, F, pairs)
        gk, gh = sd3.hiv_pair_volumes(d, P, V, F, pairs)
        assert np.array_equal(rh > 1e9, gh > 1e9) and np.array_equal(rk == 0, gk.astype(np.float32) == 0), (name, dist, rh, gh, rk, gk)
        fin = rh < 1e9
        relh = np.abs(gh[fin] - rh[fin]) / np.maximum(rh[fin], 1e-3)
        nz = rk != 0
        relk = np.abs(gk[nz] - rk[nz]) / np.maximum(rk[nz], 1e-3) if nz.any() else np.zeros(1)
        print("%s dist %.1f: hull stage max rel diff %.3g (%d finite), kernel stage %.3g (%d non-zero)" % (name, dist, relh.max(), int(fin.sum()), relk.max(), int(nz.sum())))
        assert fin.sum() >= 10 and relh.max() < 2e-6 and relk.max() < 2e-6
