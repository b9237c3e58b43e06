"""Test helper: which voxels lie exactly ON the convex hull of one of the given star polyhedra.

The one documented deviation of the 3D rasteriser: render mode "full" is  kernel OR (hull AND tetrahedra)
(stardist3d_impl.cpp:1474-1477); the reference takes the hull's planes from Qhull, normalised in double, and tests `n.p + d > 0` -- for a
voxel exactly on a hull facet or vertex that is the sign of a 1e-16 residual (of two mirror-image poles of one ellipsoid it paints one).
Lattice-aligned inputs (integer centres, integer ray lengths) put vertices and facets exactly on voxels; random float inputs never do."""
import numpy as np


def on_hull_boundary(voxels, points, dist, rays_vertices, tol=1e-6):
    from scipy.spatial import ConvexHull
    voxels = np.asarray(voxels, np.float64)
    V = np.asarray(rays_vertices, np.float32)
    on = np.zeros(len(voxels), bool)
    if not len(voxels):
        return on
    reach = np.asarray(dist, np.float32).max(axis=1) * float(np.abs(V).max()) + 2
    for c, d, r in zip(np.asarray(points, np.float32), np.asarray(dist, np.float32), reach):
        near = np.all(np.abs(voxels - c) <= r, axis=1)
        if not near.any():
            continue
        pv = (c[None] + d[:, None] * V).astype(np.float64)                 # stardist3d_impl.cpp polyhedron_polyverts (float32)
        try:
            eq = ConvexHull(pv).equations
        except Exception:                                                   # degenerate vertex set: nothing can be said
            continue
        res = voxels[near] @ eq[:, :3].T + eq[:, 3]
        idx = np.flatnonzero(near)
        on[idx] |= (res.max(axis=1) <= tol) & (np.abs(res).min(axis=1) <= tol)
    return on
