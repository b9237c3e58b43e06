"""Do the one-wave and the four-waves-per-pair exact volume routines agree on MANY pairs of the bench-like candidate sets?
(sd_hiv_pairs_device with nms3d_split_exact 0 vs 1), and which keep flags differ between the split policies on a 512^3 volume.
usage: python tools/diag_stage3x.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import _bigparity as B
from stardist_amd.lib import _native as N, stardist3d as sd3
from stardist_amd.rays3d import rays_from_json

dev = torch.device("cuda:0")
cfg = B.CFG3D_REF
model, big, axes = B.model_and_input(3, cfg, dev)
dist, prob, pts, nb = B.whole_input_candidates(model, big, axes, cfg)
rays = rays_from_json(model.config.rays_json)
V = np.ascontiguousarray(rays.vertices, np.float32); F = np.ascontiguousarray(rays.faces, np.int32)
verts = torch.as_tensor(V, device=dev); faces = torch.as_tensor(F, device=dev)
keeps = {}
for opt in (1, 2, 3, 0):
    with N.option("nms3d_split_exact", opt):
        k = sd3.c_non_max_suppression_inds(dist.float().contiguous(), pts.float().contiguous(), verts, faces, prob.float().contiguous(), 1, 1, 0, np.float32(0.3))
    keeps[opt] = k.cpu().numpy().astype(bool)
    print("split option %d: %d survivors, sha %s" % (opt, keeps[opt].sum(), B.array_digest(np.packbits(keeps[opt]))[:16]), flush=True)
for opt in (2, 3, 0):
    d = np.flatnonzero(keeps[opt] != keeps[1])
    print("option %d vs 1: %d flags differ %s" % (opt, len(d), d[:10]))
# pair volumes: close pairs among the best-scored 200k candidates of one tile
n = 200000
dd = dist[:n].cpu().numpy(); pp = pts[:n].cpu().numpy().astype(np.float32)
order = np.lexsort((pp[:, 2], pp[:, 1], pp[:, 0]))
rs = np.random.RandomState(0)
a = rs.randint(0, n - 40, 400000); b = a + rs.randint(1, 40, 400000)
pairs = np.stack([order[a], order[b]], 1).astype(np.int32)
close = np.linalg.norm(pp[pairs[:, 0]] - pp[pairs[:, 1]], axis=1) < 12
pairs = np.ascontiguousarray(pairs[close][:150000])
vols = {}
for opt in (0, 1):
    with N.option("nms3d_split_exact", opt), N.option("nms3d_volume_bounds", 0):
        vols[opt] = sd3.hiv_pair_volumes(dd, pp, V, F, pairs)
for name, k in (("kernel", 0), ("hull", 1)):
    x, y = vols[0][k], vols[1][k]
    bad = np.flatnonzero(~((x == y) | (np.isnan(x) & np.isnan(y))))
    print("%s volumes: %d pairs, %d differ between the one-wave and the four-wave routine" % (name, len(pairs), len(bad)))
    for i in bad[:8]:
        print("   pair %s: one-wave %.17g  four-wave %.17g  rel %.3g" % (pairs[i], x[i], y[i], abs(x[i] - y[i]) / max(abs(x[i]), 1e-300)))
