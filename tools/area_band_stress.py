"""Large-scale check of the 2D NMS's area enclosure against the VENDORED CLIPPER itself (oracle/_ref) on the CPU: random star-polygon pairs of
many families, numpy statement of the enclosure (tests/_area_exact.py) vs clipper_ref_area.  Prints per family the worst ratio
|A_clipper - A| / band over the usable pairs.  usage: python tools/area_band_stress.py <n_pairs_per_family> <seed> [family index ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from _area_exact import band, edge_stats, exact_area, near_pairs, near_strips, plain
from oracle import ref

FAMILIES = [  # R, radius, noise, spread of the centres, scale of the second polygon, offset of the whole pair from the origin
    (32, 10, 0.10, 12, 0.8, 50), (32, 10, 0.03, 6, 1.0, 50), (32, 10, 0.03, 3, 0.97, 50), (32, 20, 0.05, 6, 0.95, 100),
    (32, 10, 0.30, 25, 0.8, 50), (32, 6, 0.15, 8, 0.9, 30), (16, 25, 0.20, 30, 0.8, 100), (32, 40, 0.10, 60, 0.9, 200),
    (24, 100, 0.15, 150, 0.85, 500), (32, 12, 0.05, 30, 1.0, 12000), (8, 15, 0.2, 15, 0.9, 50), (32, 14, 0.02, 2, 1.0, 3000),
]


def star(rng, n, R, radius, noise, spread, off):
    ang = np.float32(2 * np.pi / R)
    k = np.arange(R, dtype=np.int32)
    s = np.sin((ang * k).astype(np.float32)).astype(np.float32); c = np.cos((ang * k).astype(np.float32)).astype(np.float32)
    d = np.maximum((radius * (1 + noise * rng.uniform(-1, 1, (n, R)))).astype(np.float32), np.float32(1e-3))
    return d, s, c


def main():
    n = int(sys.argv[1]); seed = int(sys.argv[2])
    fams = [int(a) for a in sys.argv[3:]] or list(range(len(FAMILIES)))
    for fi in fams:
        R, radius, noise, spread, scale, off = FAMILIES[fi]
        rng = np.random.RandomState(seed * 1000 + fi)
        worst = 0.0; worst0 = 0.0; nus = 0; ntot = 0; t0 = time.time(); und = {0.3: 0, 0.4: 0, 0.5: 0}
        for s0 in range(0, n, 20000):
            m = min(20000, n - s0)
            pa = np.floor(rng.uniform(off, off + spread, (m, 2))).astype(np.float32); pb = np.floor(rng.uniform(off, off + spread, (m, 2))).astype(np.float32)
            da, sn, cs = star(rng, m, R, radius, noise, spread, off); db, _, _ = star(rng, m, R, radius * scale, noise, spread, off)
            xa = (pa[:, 1:] + da * cs).astype(np.float32).astype(np.int64); ya = (pa[:, :1] + da * sn).astype(np.float32).astype(np.int64)
            xb = (pb[:, 1:] + db * cs).astype(np.float32).astype(np.int64); yb = (pb[:, :1] + db * sn).astype(np.float32).astype(np.int64)
            A, K, ok, _, _ = exact_area(xa, ya, xb, yb)
            us = ok & plain(xa, ya) & plain(xb, yb)
            la, pea = edge_stats(xa, ya); lb, peb = edge_stats(xb, yb)
            ext = np.maximum(np.abs(xa - xa.mean(1, keepdims=True)).max(1), np.abs(xb - xa.mean(1, keepdims=True)).max(1)) + radius
            B = band(K, near_pairs(xa, ya, xb, yb), la, lb, ext, pea, peb, near_strips(xa, ya, xb, yb))
            C = np.array([ref.clipper_area(xa[i], ya[i], xb[i], yb[i]) for i in range(m)], np.float64)
            d = np.abs(C - A)
            if us.any():
                worst = max(worst, float((d[us] / B[us]).max()))
                z = us & (K == 0)
                if z.any(): worst0 = max(worst0, float(d[z].max()))
                amin = np.minimum(np.abs((xa * np.roll(ya, -1, 1) - ya * np.roll(xa, -1, 1)).sum(1)), np.abs((xb * np.roll(yb, -1, 1) - yb * np.roll(xb, -1, 1)).sum(1))) / 2 + 1e-10
                for thr in und: und[thr] += int((~us | (np.abs(A / amin - thr) <= B / amin)).sum())
            nus += int(us.sum()); ntot += m
        print("family %2d R=%d radius=%g noise=%g spread=%g scale=%g offset=%g: %d pairs, usable %.4f, max |A_clipper - A| / band %.3f, max |d| at K=0 %.2f, undecided at thr 0.3/0.4/0.5: %s  (%.0f s)"
              % (fi, R, radius, noise, spread, scale, off, ntot, nus / max(1, ntot), worst, worst0, "/".join("%.3f" % (und[t] / ntot) for t in (0.3, 0.4, 0.5)), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
