"""Rays_Cartesian, lattice family "int": which stage makes the device NMS leave the reference?  (a) result-preserving options switched one at
a time, (b) for the first candidates whose flag differs: every pair (kept i, j) through the reference's cascade (oracle shim) and through the
device routines (kernel / hull volumes, rendered overlap from sd_inside_polyhedron_device).  GPU box; test infrastructure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from make_lattice_golden import rays_of
from oracle import synth, ref
from stardist_amd.lib import stardist3d as sd3, _native as N

dev = torch.device("cuda:0")
G = dict(np.load(os.path.join(ROOT, "tests", "golden", "lattice_reference.npz"))); G.update(dict(np.load(os.path.join(ROOT, "tests", "golden", "lattice_reference_more.npz"))))
NAME = sys.argv[2] if len(sys.argv) > 2 else "cartesian_8_5"
from make_lattice_golden_more import RAYS3D_MORE, rays_of_more
rays = rays_of_more(NAME) if NAME in RAYS3D_MORE else rays_of(NAME)
V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
tV, tF = torch.from_numpy(V).to(dev), torch.from_numpy(F).to(dev)
fam = sys.argv[1] if len(sys.argv) > 1 else "int"
d, p, s = synth.lattice_candidates_3d(len(V), fam, size=48)
P = p.astype(np.float32); n = len(d)

def inside(k, pts, use_map=1):
    tp = torch.from_numpy(np.ascontiguousarray(pts, np.float32)).to(dev)
    o = torch.empty(len(pts), dtype=torch.uint8, device=dev)
    td = torch.from_numpy(np.ascontiguousarray(d[k])).to(dev); tc = torch.from_numpy(np.ascontiguousarray(P[k])).to(dev)     # (kept alive over the call)
    N.dcall(tp, "sd_inside_polyhedron_device", N.tptr(td), N.tptr(tc), len(V), len(F), N.tptr(tV), N.tptr(tF), N.tptr(tp), len(pts), use_map, N.tptr(o))
    r = o.cpu().numpy().astype(bool)
    del td, tc
    return r

for thr in (0.2, 0.4):
    want = np.unpackbits(G["nms3d_%s_%s_%.1f" % (NAME, fam, thr)])[:n].astype(bool)
    keep = np.asarray(sd3.c_non_max_suppression_inds(d, P, V, F, s, 1, 1, 0, np.float32(thr))).astype(bool)
    diff = np.flatnonzero(keep != want)
    print("RESULT", fam, thr, "default options: flags differ on", len(diff), diff[:12].tolist())
    for opt, val in (("nms3d_cone_map", 0), ("nms3d_volume_bounds", 0), ("nms3d_tail_batch", 0), ("nms3d_split_exact", 0), ("nms3d_refine_mesh", 0)):
        with N.option(opt, val):
            k2 = np.asarray(sd3.c_non_max_suppression_inds(d, P, V, F, s, 1, 1, 0, np.float32(thr))).astype(bool)
        print("RESULT   %s = %d: flags differ on %d" % (opt, val, int((k2 != want).sum())))
    for j in diff[:6]:
        cands = [i for i in range(j) if want[i] and np.abs(P[i] - P[j]).max() < 16]
        if not cands: print("RESULT   candidate", j, "no kept neighbour before it"); continue
        pairs = np.array([[i, j] for i in cands], np.int32)
        C = ref.pair_cascade(d, P, V, F, pairs)
        gk, gh = sd3.hiv_pair_volumes(d, P, V, F, pairs)
        print("RESULT   candidate %d: reference keeps %s, device keeps %s" % (j, bool(want[j]), bool(keep[j])))
        for (i, _), c, k_, h_ in zip(pairs, C, np.asarray(gk), np.asarray(gh)):
            amin = min(c[0], c[1]) + 1e-10
            if c[2] < 1e-10 or c[2] / amin <= thr: continue
            # device rendered overlap: voxels of i's box inside both
            lo = np.floor(P[i] - d[i].max() - 1).astype(int); hi = np.ceil(P[i] + d[i].max() + 1).astype(int)
            g = np.stack(np.meshgrid(*[np.arange(lo[a], hi[a] + 1) for a in range(3)], indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
            ov = [int((inside(i, g, m) & inside(j, g, m)).sum()) for m in (0, 1)]
            print("RESULT     i = %d: volumes %.3f %.3f | kernel ref %.4g dev %.4g | hull ref %.6g dev %.6g (ratio to the smaller volume %.4f / %.4f) | rendered overlap ref %d dev %d (full loop) %d (cone map) -> ratio %.4f"
                  % (i, c[0], c[1], c[4], k_, c[5], h_, c[5] / amin, h_ / amin, int(c[6]), ov[0], ov[1], c[6] / amin))
