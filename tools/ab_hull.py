"""A/B of the convex-hull construction of the 3D NMS (stage 4): hull volumes of every polyhedron of a candidate set (pairs (i, i) and
(i, i + 1) through sd_hiv_pairs_device) as a SHA-256, and the time of the hull construction alone (a call with one pair builds
all N hulls).  usage: python tools/ab_hull.py <path of libstardist_hip.so> [size]   (run once per library, compare the lines)"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import synth
from stardist_amd.lib import _native
_native.LIB_PATH = sys.argv[1]
from stardist_amd.lib import stardist3d as sd3
from stardist_amd.rays3d import Rays_GoldenSpiral

size = int(sys.argv[2]) if len(sys.argv) > 2 else 160
for R in (96, 32):
    rays = Rays_GoldenSpiral(R)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    d, p, s, nobj = synth.s3d_nuclei(size, V)
    for noise in (0.0, 0.05):
        dd = (d * (1 + noise * np.random.RandomState(1).standard_normal(d.shape))).astype(np.float32) if noise else d
        n = len(dd)
        pairs = np.concatenate([np.stack([np.arange(n), np.arange(n)], 1), np.stack([np.arange(n - 1), np.arange(1, n)], 1)]).astype(np.int32)
        _, vh = sd3.hiv_pair_volumes(dd, p, V, F, pairs, kernel=False, hull=True)
        h = hashlib.sha256(vh.tobytes()).hexdigest()[:16]
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t = time.time()
            sd3.hiv_pair_volumes(dd, p, V, F, pairs[:1], kernel=False, hull=True)
            ts.append(time.time() - t)
        print(f"R={R} noise={noise}: N={n} hull volumes sha {h}  failed {int((vh >= 1e9).sum())}  mean vol {vh[:n][vh[:n] < 1e9].mean():.3f}  "
              f"all-hulls call {min(ts) * 1e3:.2f} ms (incl. upload)", flush=True)
