"""CPU: the inscribed / circumscribed polytope bounds used by the 3D NMS (nms3d.hip::hiv_bounds_wave) are rigorous.
numpy restatement of the bound construction, checked against exact half-space-intersection volumes from scipy's Qhull
(independent of the product code and of the oracle build)."""
import numpy as np
import pytest
from scipy.spatial import ConvexHull, HalfspaceIntersection


def _halfspaces(center, dist, rays_v, faces):
    """kernel half-spaces of a star-convex polyhedron: inside <=> n.p + d <= 0 (one per face, oriented by the centre)"""
    P = center[None] + dist[:, None] * rays_v
    A, B, C = P[faces[:, 0]], P[faces[:, 1]], P[faces[:, 2]]
    n = np.cross(B - A, C - A)
    d = -np.einsum("ij,ij->i", n, A)
    flip = (n @ center + d) > 0
    n[flip] *= -1; d[flip] *= -1
    nn = np.linalg.norm(n, axis=1)
    ok = nn > 0
    return n[ok] / nn[ok, None], d[ok] / nn[ok]


def _bounds(n, d, c, dirs, faces):
    """lower / upper bound of vol{x: n.x + d <= 0} from one ray cast per direction from the interior point c"""
    e = n @ c + d                      # < 0
    q = n @ dirs.T                     # (M, R)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(q > 0, -e[:, None] / q, np.inf)
    hit = np.argmin(t, axis=0)
    tt = t[hit, np.arange(dirs.shape[0])]
    if not np.all(np.isfinite(tt)):
        return None
    w = tt[:, None] * dirs             # boundary points relative to c
    wa, wb, wc = w[faces[:, 0]], w[faces[:, 1]], w[faces[:, 2]]
    det = np.abs(np.einsum("ij,ij->i", wa, np.cross(wb, wc)))
    lb = det.sum() / 6.0
    best = np.full(len(faces), np.inf)
    for x in range(3):
        m = hit[faces[:, x]]
        ne = -e[m]
        prod = np.ones(len(faces)); ok = np.ones(len(faces), bool)
        for y in (wa, wb, wc):
            qq = np.einsum("ij,ij->i", n[m], y)
            ok &= qq > 0
            prod *= ne / np.where(qq > 0, qq, 1.0)
        best = np.where(ok, np.minimum(best, prod), best)
    if not np.all(np.isfinite(best)):
        return lb, np.inf
    return lb, float((det * np.maximum(best, 1.0)).sum() / 6.0)


@pytest.mark.parametrize("n_rays,seed", [(32, 0), (96, 1), (96, 2)])
def test_volume_bounds_are_rigorous_and_tight(n_rays, seed):
    from stardist_amd.rays3d import Rays_GoldenSpiral
    rays = Rays_GoldenSpiral(n_rays)
    V, F = rays.vertices.astype(np.float64), rays.faces.astype(np.int64)
    rng = np.random.RandomState(seed)
    ratios = []
    for _ in range(60):
        r1, r2 = rng.uniform(6, 10, 2)
        d1 = r1 * (1 + 0.08 * rng.uniform(-1, 1, n_rays)); d2 = r2 * (1 + 0.08 * rng.uniform(-1, 1, n_rays))
        c1 = rng.uniform(20, 30, 3); c2 = c1 + rng.uniform(-1, 1, 3) * rng.uniform(0.5, 9)
        n1, o1 = _halfspaces(c1, d1, V, F); n2, o2 = _halfspaces(c2, d2, V, F)
        n, d = np.concatenate([n1, n2]), np.concatenate([o1, o2])
        c = 0.5 * (c1 + c2)
        if np.any(n @ c + d >= -1e-9):
            continue                                        # interior point infeasible: the reference treats this as volume 0
        hs = HalfspaceIntersection(np.concatenate([n, d[:, None]], 1), c)
        vol = ConvexHull(hs.intersections).volume
        b = _bounds(n, d, c, V, F)
        assert b is not None
        lb, ub = b
        assert lb <= vol * (1 + 1e-9), (lb, vol)
        assert ub >= vol * (1 - 1e-9), (ub, vol)
        ratios.append((lb / vol, ub / vol))
    ratios = np.array(ratios)
    assert len(ratios) >= 30
    # tightness (informative: the bounds only need to be valid): typical gap for rough random polyhedra
    print("n_rays=%d: median lower/exact %.3f, median upper/exact %.3f" % (n_rays, np.median(ratios[:, 0]), np.median(ratios[:, 1])))
    assert np.median(ratios[:, 0]) > 0.6 and np.median(ratios[:, 1]) < 1.6
