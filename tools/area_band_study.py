"""CPU study for a decision shortcut of the 2D NMS: the exact area of the intersection of two integer star polygons by boundary
integration (regular O(n^2) arithmetic, exact integer predicates with a symbolic perturbation) against Clipper's area (the vendored
Clipper through oracle/_ref: rounds crossing points to the lattice).  Reports the deviation A_clipper - A_exact against the number of
boundary crossings, and which share of realistic candidate pairs a band around the threshold would leave to the exact sweep.
usage: python tools/area_band_study.py [n_objects] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref


sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from _area_exact import exact_area, plain


def candidates(n_side, seed, n_rays=32, spacing=32, R=(8, 14), noise=0.03, frac=0.45):
    rng = np.random.RandomState(seed)
    g = np.arange(spacing // 2, n_side * spacing, spacing)
    cy, cx = np.meshgrid(g, g, indexing="ij")
    C = np.stack([cy.ravel(), cx.ravel()], 1).astype(np.float64) + rng.uniform(-6, 6, (len(g) ** 2, 2))
    Rs = rng.uniform(R[0], R[1], len(C))
    ang = (2 * np.pi / n_rays * np.arange(n_rays)).astype(np.float32)      # stardist2d.cpp:438-441: float angles
    sn, cs = np.sin(ang.astype(np.float64)).astype(np.float32), np.cos(ang.astype(np.float64)).astype(np.float32)
    P, D, O = [], [], []
    for k, (c, r) in enumerate(zip(C, Rs)):
        rc = int(frac * r)
        o = np.arange(-rc, rc + 1)
        off = np.stack(np.meshgrid(o, o, indexing="ij"), -1).reshape(-1, 2)
        off = off[(off ** 2).sum(1) <= rc * rc]
        p = np.round(c).astype(np.int64) + off
        q = p - c
        bq = q[:, :1] * sn[None].astype(np.float64) + q[:, 1:] * cs[None].astype(np.float64)
        t = -bq + np.sqrt(bq * bq - ((q * q).sum(1)[:, None] - r * r))
        t *= 1 + noise * rng.standard_normal(t.shape)
        P.append(p); D.append(np.maximum(t, 1e-3)); O.append(np.full(len(p), k))
    P = np.concatenate(P).astype(np.float32); D = np.concatenate(D).astype(np.float32); O = np.concatenate(O)
    y = P[:, :1] + D * sn[None]; x = P[:, 1:] + D * cs[None]                  # float32, no fma (:454-455)
    return P, D, O, x.astype(np.int64), y.astype(np.int64), C, Rs


def main():
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    P, D, O, VX, VY, C, Rs = candidates(n_side, seed)
    N = len(P)
    rng = np.random.RandomState(seed + 1)
    # pairs: same object (sample) + neighbouring objects
    from scipy.spatial import cKDTree
    tree = cKDTree(P)
    rad = D.max(1)
    pairs = []
    for i in rng.choice(N, min(N, 6000), replace=False):
        nb = np.array(tree.query_ball_point(P[i], rad[i] + 15.0))
        nb = nb[nb != i]
        if len(nb) > 40: nb = rng.choice(nb, 40, replace=False)
        pairs += [(i, j) for j in nb]
    pairs = np.array(pairs)
    print(f"{N} candidates of {len(C)} objects, {len(pairs)} pairs", flush=True)
    t0 = time.time()
    pl = plain(VX, VY)
    print(f"plain polygons: {pl.mean():.4f}  ({time.time() - t0:.1f} s)", flush=True)
    area2 = np.abs((VX * np.roll(VY, -1, 1) - VY * np.roll(VX, -1, 1)).sum(1)) * 0.5
    A_c = np.empty(len(pairs)); t0 = time.time()
    for k, (i, j) in enumerate(pairs):
        A_c[k] = ref.clipper_area(VX[i], VY[i], VX[j], VY[j])
    print(f"clipper: {time.time() - t0:.1f} s", flush=True)
    A_e = np.empty(len(pairs)); K = np.empty(len(pairs), int); OK = np.empty(len(pairs), bool)
    t0 = time.time()
    for s in range(0, len(pairs), 4000):
        ii, jj = pairs[s:s + 4000, 0], pairs[s:s + 4000, 1]
        A_e[s:s + 4000], K[s:s + 4000], OK[s:s + 4000], _, _ = exact_area(VX[ii], VY[ii], VX[jj], VY[jj])
    print(f"exact: {time.time() - t0:.1f} s", flush=True)
    good = OK & pl[pairs[:, 0]] & pl[pairs[:, 1]]
    d = A_c - A_e
    amin = np.minimum(area2[pairs[:, 0]], area2[pairs[:, 1]]) + 1e-10
    print(f"pairs with both polygons plain and equally oriented: {good.mean():.4f}")
    for kk in sorted(set(K[good])):
        m = good & (K == kk)
        print(f"  K={kk:3d}: {m.sum():7d} pairs  d = A_clipper - A_exact: mean {d[m].mean():+.3f}  min {d[m].min():+.3f}  max {d[m].max():+.3f}   max|d|/K {np.abs(d[m]).max() / max(kk, 1):.3f}")
    m = good
    print(f"all good pairs: max |d| {np.abs(d[m]).max():.3f}, max |d|/max(K,1) {(np.abs(d[m]) / np.maximum(K[m], 1)).max():.3f}; bad pairs max|d| {np.abs(d[~m]).max() if (~m).any() else 0:.3f}")
    iou = A_e / amin
    for thr in (0.3, 0.4, 0.5):
        for per in (1.0, 1.5, 2.0, 3.0):
            band = (per * np.maximum(K, 1) + 0.5) / amin
            und = ~good | (np.abs(iou - thr) <= band)
            wrong = good & ~und & ((A_c / amin > thr) != (iou > thr))
            print(f"  thr {thr}: band {per} px^2 per crossing -> undecided {und.mean():.4f}, wrong decisions {wrong.sum()}")
    np.savez("/tmp/area_band_study.npz", d=d, K=K, good=good, iou=iou, A_c=A_c, A_e=A_e, amin=amin)


if __name__ == "__main__":
    main()
