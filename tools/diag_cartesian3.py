"""hull-stage volumes of shifted copies of ONE polyhedron (Rays_Cartesian and others, constant distance) against scipy's Qhull on the box"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
from scipy.spatial import ConvexHull, HalfspaceIntersection
from make_lattice_golden import rays_of
from stardist_amd.lib import stardist3d as sd3
shifts = [(0, 0, 0), (1e-5, 0, 0), (1e-5, 2e-5, 3e-5), (1, 0, 0), (1.00001, 0, 0), (1, 1e-5, 0), (1, 1e-5, 2e-5), (2, 0, 0), (0, 1, 0), (1, 1, 0), (1, 1.00001, 0), (3, 0, 0), (0.5, 0, 0)]
for name in ("cartesian_8_5", "octo", "golden32"):
    rays = rays_of(name)
    V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
    for dist in (5.0,):
        c0 = np.array([24, 24, 24], np.float32)
        P = np.array([c0] + [c0 + np.array(sh, np.float32) for sh in shifts], np.float32)
        d = np.full((len(P), len(V)), dist, np.float32)
        pairs = np.array([[0, k + 1] for k in range(len(shifts))], np.int32)
        gk, gh = sd3.hiv_pair_volumes(d, P, V, F, pairs)
        for (i, j), h_ in zip(pairs, np.asarray(gh)):
            pv1 = (P[i][None] + d[i][:, None] * V).astype(np.float32).astype(np.float64); pv2 = (P[j][None] + d[j][:, None] * V).astype(np.float32).astype(np.float64)
            h1, h2 = ConvexHull(np.unique(pv1, axis=0)), ConvexHull(np.unique(pv2, axis=0))
            hs = np.concatenate([h1.equations, h2.equations]); mid = 0.5 * (P[i].astype(np.float64) + P[j])
            try: vol = ConvexHull(HalfspaceIntersection(hs, mid).intersections).volume
            except Exception as e: vol = float("nan")
            flag = "" if abs(vol - h_) <= 1e-6 * max(vol, 1) else "   <-- differs"
            # kernels: the half-spaces of the mesh faces (build_halfspace, stardist3d_impl.cpp:744-764), float32 vertices
            def khs(pv):
                A, B, C = pv[F[:, 0]], pv[F[:, 1]], pv[F[:, 2]]
                Pq, Q = B - A, C - A
                Nn = -np.stack([Pq[:, 1] * Q[:, 2] - Pq[:, 2] * Q[:, 1], Pq[:, 2] * Q[:, 0] - Pq[:, 0] * Q[:, 2], Pq[:, 0] * Q[:, 1] - Pq[:, 1] * Q[:, 0]], 1)
                Nn = np.stack([-(Pq[:, 1] * Q[:, 2] - Pq[:, 2] * Q[:, 1]), -(Pq[:, 2] * Q[:, 0] - Pq[:, 0] * Q[:, 2]), -(Pq[:, 0] * Q[:, 1] - Pq[:, 1] * Q[:, 0])], 1)
                return np.concatenate([Nn, -(A * Nn).sum(1, keepdims=True)], 1)
            try:
                hk = np.concatenate([khs(pv1), khs(pv2)]); hk = hk[np.abs(hk[:, :3]).sum(1) > 0]
                sgn = np.sign(-(hk[:, :3] @ mid + hk[:, 3]).min())
                kv = ConvexHull(HalfspaceIntersection(hk if (hk[:, :3] @ mid + hk[:, 3]).max() < 0 else -hk, mid).intersections).volume
            except Exception as e: kv = float("nan")
            print("RESULT %s dist %.1f shift %s: hull stage device %.6f scipy %.6f%s | kernel stage device %.6f scipy %.6f" % (name, dist, shifts[j - 1], h_, vol, flag, gk[j - 1], kv))
