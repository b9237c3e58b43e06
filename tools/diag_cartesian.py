"""Rays_Cartesian on the lattice sets: where does the 3D NMS leave the reference?  Pair volumes of both Qhull
stages against the device routines, stage counters of both NMS runs, the candidates whose keep flag differs.  GPU box; test infrastructure
(reads the compiled reference under oracle/_ref).  usage: python tools/diag_cartesian.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
from make_lattice_golden import rays_of
from oracle import synth, ref
from stardist_amd.lib import stardist3d as sd3

rays = rays_of("cartesian_8_5")
V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
for fam in ("const", "int"):
    d, p, s = synth.lattice_candidates_3d(len(V), fam, size=48)
    n = len(d); P = p.astype(np.float32)
    ii, jj = np.triu_indices(n, 1)
    sep = np.sqrt(((P[ii] - P[jj]) ** 2).sum(1))
    sel = np.flatnonzero(sep < 12)[:3000]
    pairs = np.stack([ii[sel], jj[sel]], 1).astype(np.int32)
    rk, rh = ref.pair_volumes(d, P, V, F, pairs)
    gk, gh = sd3.hiv_pair_volumes(d, P, V, F, pairs)
    gk = np.asarray(gk); gh = np.asarray(gh)
    print("RESULT", fam, "pairs", len(pairs), "| kernel == 0: reference", int((rk == 0).sum()), "device", int((gk.astype(np.float32) == 0).sum()),
          "| hull error value: reference", int((rh > 1e9).sum()), "device", int((gh > 1e9).sum()), "pattern differs on", int(((rh > 1e9) != (gh > 1e9)).sum()))
    both = (rh < 1e9) & (gh < 1e9)
    rel = np.abs(gh[both] - rh[both]) / np.maximum(np.abs(rh[both]), 1e-3)
    print("RESULT   hull volumes finite in both:", int(both.sum()), "max rel diff %.3g" % (rel.max() if both.any() else 0), "pairs above 1e-5:", int((rel > 1e-5).sum()))
    bad = np.flatnonzero((rh > 1e9) != (gh > 1e9))[:12]
    for b in bad:
        i, j = pairs[b]
        print("RESULT   pair", int(i), int(j), "centres", P[i].tolist(), P[j].tolist(), "dist", np.unique(d[i]).tolist(), np.unique(d[j]).tolist(), "reference", float(rh[b]), "device", float(gh[b]))
    worst = np.argsort(-rel)[:6] if both.any() else []
    idx = np.flatnonzero(both)
    for w in worst:
        b = idx[w]; i, j = pairs[b]
        print("RESULT   volume pair", int(i), int(j), "centres", P[i].tolist(), P[j].tolist(), "reference %.6f device %.6f" % (rh[b], gh[b]))
    o = np.argsort(s, kind="stable")[::-1]
    dd, pp, ss = np.ascontiguousarray(d[o]), np.ascontiguousarray(P[o]), np.ascontiguousarray(s[o])
    for thr in (0.2, 0.4):
        sys.stdout.flush()
        print("RESULT -- reference NMS, threshold", thr); sys.stdout.flush()
        want = np.asarray(ref.stardist3d().c_non_max_suppression_inds(dd, pp, V, F, ss, 1, 1, 1, np.float32(thr))).astype(bool)
        sys.stdout.flush()
        print("RESULT -- device NMS, threshold", thr); sys.stdout.flush()
        keep = np.asarray(sd3.c_non_max_suppression_inds(dd, pp, V, F, ss, 1, 1, 1, np.float32(thr))).astype(bool)
        diff = np.flatnonzero(keep != want)
        print("RESULT   keep flags differ on", len(diff), "of", n, ":", diff[:20].tolist(), "(reference keeps", int(want.sum()), ", device", int(keep.sum()), ")")
