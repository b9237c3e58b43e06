"""Diagnostic (GPU box): where relabel_image_stardist3D differs from the reference's golden for the anisotropic ball -- star_dist3D or the rasteriser."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import port, ref
from stardist_amd import Rays_GoldenSpiral
from stardist_amd.geometry import geom3d
from stardist_amd.geometry.geom2d import _region_centroids
G = np.load(os.path.join(ROOT, "tests", "golden", "relabel_reference.npz"))
ref.set_threads(1)
for k in range(int(G["n3d"])):
    lbl = G["in3d_%d" % k]
    rays = Rays_GoldenSpiral(int(G["rays3d_%d" % k]), anisotropy=tuple(1.0 / G["eps3d_%d" % k]))
    d_hip = np.asarray(geom3d.star_dist3D(lbl, rays))
    d_ref = port.star_dist3D(lbl, rays.vertices)
    nd = int((d_hip != d_ref).sum())
    labs, cen = _region_centroids(lbl)
    pts = cen.astype(int)
    dist = np.maximum(d_ref[tuple(pts.T)].reshape(len(pts), len(rays)), 1e-3)
    dist_h = np.maximum(d_hip[tuple(pts.T)].reshape(len(pts), len(rays)), 1e-3)
    r_hip = np.asarray(geom3d.polyhedron_to_label(dist, pts, rays, shape=lbl.shape, labels=labs, verbose=False)).astype(np.int32)
    r_ref = port.polyhedron_to_label(dist, pts, rays.vertices, rays.faces, lbl.shape, labels=labs, verbose=False).astype(np.int32)
    diff = np.argwhere(r_hip != r_ref)
    print("case", k, str(G["name3d_%d" % k]), "rays", len(rays), "| star_dist3D values differing:", nd, "of", d_ref.size,
          "max |diff|", float(np.abs(d_hip - d_ref).max()), "| dist rows at centres equal:", bool(np.array_equal(dist, dist_h)),
          "| raster voxels differing (same dist):", len(diff), "| ref == golden:", bool(np.array_equal(r_ref, G["out3d_%d" % k])))
    for p in diff[:12]:
        c = pts[0]
        print("   voxel", tuple(int(v) for v in p), "hip", int(r_hip[tuple(p)]), "ref", int(r_ref[tuple(p)]), "offset from centre", tuple(int(a - b) for a, b in zip(p, c)))
    if nd:
        w = np.argwhere(d_hip != d_ref)[:6]
        for q in w:
            print("   dist", tuple(int(v) for v in q), "hip", float(d_hip[tuple(q)]), "ref", float(d_ref[tuple(q)]))
