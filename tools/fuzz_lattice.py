"""Lattice fuzz (GPU box): the natives against the compiled reference on inputs whose geometry sits EXACTLY on the pixel lattice --
integer centres, integer distances, few rays (squares, diamonds, octagons), identical and one-pixel-shifted shapes -- the tie cases
random float inputs never produce (coincident edges and vertices for Clipper and the area band, voxels exactly on faces for the 3D
predicates, polygon vertices on pixel centres for the rasterisers).  Prints one line per configuration: number of differing keep flags /
label pixels.  usage: python tools/fuzz_lattice.py [seeds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cands2d(rng, H, W, R, n, family):
    pts = np.stack([rng.randint(2, H - 2, n), rng.randint(2, W - 2, n)], 1).astype(np.float32)
    if family == "const":                         # regular R-gons of integer radius (R = 4: diamonds whose vertices are lattice points)
        d = np.repeat(rng.randint(2, 9, (n, 1)), R, 1)
    elif family == "int":                         # every ray its own integer length
        d = rng.randint(2, 9, (n, R))
    else:                                         # half-integers
        d = rng.randint(4, 18, (n, R)) * 0.5
    s = rng.uniform(0, 1, n).astype(np.float32)
    ind = np.argsort(s, kind="stable")[::-1]
    return np.ascontiguousarray(d[ind].astype(np.float32)), np.ascontiguousarray(pts[ind]), np.ascontiguousarray(s[ind])


def main():
    from oracle import port, ref
    from stardist_amd.geometry import polygons_to_label, polyhedron_to_label
    from stardist_amd.lib import _native, stardist2d as sd2, stardist3d as sd3
    from stardist_amd.rays3d import Rays_Cartesian, Rays_GoldenSpiral, Rays_Octo
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    m2, m3 = ref.stardist2d(), ref.stardist3d()
    ref.set_threads(1)
    t0 = time.time()
    bad = 0
    for R in (4, 8, 16, 32):
        for family in ("const", "int", "half"):
            for seed in range(seeds):
                rng = np.random.RandomState(1000 * R + seed)
                d, p, s = cands2d(rng, 96, 96, R, 1800, family)
                for thr in (0.3, 0.5):
                    rk = m2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr)).astype(bool)
                    k = sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr))
                    with _native.option("nms2d_strict", 1):
                        ks = sd2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr))
                    nd, ns = int((k != rk).sum()), int((ks != rk).sum())
                    bad += nd + ns
                    print("nms2d R=%2d %-5s seed %d thr %.1f: %4d kept | flags differing: default %d, strict %d" % (R, family, seed, thr, int(rk.sum()), nd, ns), flush=True)
                keep = m2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.3)).astype(bool)
                a = np.asarray(polygons_to_label(d[keep], p[keep], (96, 96), prob=s[keep]))
                b = port.polygons_to_label(d[keep], p[keep], (96, 96), prob=s[keep])
                nd = int((a != b).sum())
                bad += nd
                print("raster2d R=%2d %-5s seed %d: %d polygons | pixels differing from the skimage restatement: %d" % (R, family, seed, int(keep.sum()), nd), flush=True)
    for name, rays in (("cartesian", Rays_Cartesian(8, 5)), ("octo", Rays_Octo()), ("golden32", Rays_GoldenSpiral(32)), ("golden32_aniso", Rays_GoldenSpiral(32, anisotropy=(2, 1, 1)))):
        V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
        for family in ("const", "int"):
            rng = np.random.RandomState(len(V) * 7 + (family == "int"))
            n = 500
            pts = np.stack([rng.randint(4, 44, n) for _ in range(3)], 1).astype(np.float32)
            d = (np.repeat(rng.randint(3, 8, (n, 1)), len(V), 1) if family == "const" else rng.randint(3, 8, (n, len(V)))).astype(np.float32)
            s = rng.uniform(0, 1, n).astype(np.float32)
            ind = np.argsort(s, kind="stable")[::-1]
            d, pts, s = np.ascontiguousarray(d[ind]), np.ascontiguousarray(pts[ind]), np.ascontiguousarray(s[ind])
            for thr in (0.2, 0.4):
                rk = m3.c_non_max_suppression_inds(d, pts, V, F, s, 1, 1, 0, np.float32(thr)).astype(bool)
                k, st = sd3.c_non_max_suppression_inds(d, pts, V, F, s, 1, 1, 0, np.float32(thr), return_stats=True)
                nd = int((k != rk).sum())
                bad += nd
                print("nms3d %-14s %-5s thr %.1f: %4d kept of %d | flags differing: %d | near-threshold exact volumes: %d" % (name, family, thr, int(rk.sum()), n, nd, int(st[13])), flush=True)
            keep = m3.c_non_max_suppression_inds(d, pts, V, F, s, 1, 1, 0, np.float32(0.2)).astype(bool)
            lab = np.arange(1, keep.sum() + 1, dtype=np.int32)
            for mode, mname in ((0, "full"), (1, "kernel")):
                a = np.asarray(sd3.c_polyhedron_to_label(d[keep], pts[keep], V, F, lab, np.int32(mode), np.int32(0), np.int32(0), np.int32(0), (48, 48, 48)))
                b = m3.c_polyhedron_to_label(d[keep], pts[keep], V, F, lab, mode, 0, 0, 0, (48, 48, 48))
                nd = int((a != b).sum())
                print("raster3d %-14s %-5s mode %-6s: %d polyhedra | voxels differing: %d (painted by us only: %d, by the reference only: %d)"
                      % (name, family, mname, int(keep.sum()), nd, int(((a != b) & (b == 0)).sum()), int(((a != b) & (a == 0)).sum())), flush=True)
    print("total differing flags / 2D pixels: %d   (%.1f s)" % (bad, time.time() - t0))


if __name__ == "__main__":
    main()
