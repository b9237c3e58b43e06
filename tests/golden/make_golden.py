"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref, built from /root/reference).

Run in the build container (where /root/reference exists):  python tests/golden/make_golden.py
The fixtures pin (a) the oracle itself across rebuilds and (b) give the GPU box known answers that
do not depend on oracle/_ref having travelled.  Inputs are regenerated from seeds by the tests
(oracle/synth.py); only outputs (and tiny inputs) are stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import port, ref, synth  # noqa: E402
from stardist_amd.rays3d import Rays_GoldenSpiral  # noqa: E402

out = {}
m2, m3 = ref.stardist2d(), ref.stardist3d()
ref.set_threads(1)

# 2D NMS (survey calibration points: 512^2 -> 26010 candidates, 1622 survivors)
for (H, W, R, thr) in [(256, 256, 32, 0.4), (512, 512, 32, 0.4), (356, 299, 11, 0.5), (114, 217, 32, 0.3)]:
    d, p, s = synth.s2d_uniform(H, W, n_rays=R)
    keep = m2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr))
    out["nms2d_%d_%d_%d_keep" % (H, W, R)] = np.packbits(keep)
    out["nms2d_%d_%d_%d_n" % (H, W, R)] = np.array([len(d), int(keep.sum())])

# star_dist 2D
lbl, _, _ = synth.s2d_nuclei_labels(200, 231, seed=3)
for R, g in [(32, (1, 1)), (17, (2, 2))]:
    out["stardist2d_%d_%d" % (R, g[0])] = m2.c_star_dist(lbl, R, g[0], g[1]).astype(np.float16 if False else np.float32)[::7, ::7]

# 3D
rays = Rays_GoldenSpiral(96)
V, F = rays.vertices, rays.faces.astype(np.int32)
d, p, s, nobj = synth.s3d_nuclei(64, V)
keep = m3.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(0.3))
out["nms3d_64_keep"] = np.packbits(keep)
out["nms3d_64_n"] = np.array([len(d), int(keep.sum()), nobj])
lbl3 = m3.c_polyhedron_to_label(d[keep], p[keep], V, F, np.arange(1, keep.sum() + 1, dtype=np.int32), 0, 0, 0, 0, (64, 64, 64))
out["raster3d_64_hist"] = np.bincount(lbl3.ravel())
out["raster3d_64_sum"] = np.array([int((lbl3.astype(np.int64) * np.arange(lbl3.size).reshape(lbl3.shape) % 1000003).sum())])

# clipper pair areas on a fixed tiny set
rng = np.random.RandomState(7)
xa = rng.randint(40, 70, (64, 12)); ya = rng.randint(40, 70, (64, 12)); xb = rng.randint(40, 70, (64, 12)); yb = rng.randint(40, 70, (64, 12))
out["clip_xa"], out["clip_ya"], out["clip_xb"], out["clip_yb"] = xa.astype(np.int16), ya.astype(np.int16), xb.astype(np.int16), yb.astype(np.int16)
out["clip_area"] = np.array([ref.clipper_area(xa[i], ya[i], xb[i], yb[i]) for i in range(64)], np.float32)

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "reference_outputs.npz"), **out)
print({k: (v.shape, v.dtype) for k, v in out.items()})
