"""Golden survivor bits of the compiled reference at the sizes BASELINE.json's metric is quoted on (TEST INFRASTRUCTURE).

  nms3d_256: S3D-nuclei 256^3 (150 606 candidates -> 1 328 survivors), Rays_GoldenSpiral(96), threshold 0.3,
             stardist.lib.stardist3d.c_non_max_suppression_inds of the reference compiled by oracle/Makefile, ONE OpenMP
             thread (the reference's anisotropy accumulation is only defined for one thread, stardist3d_impl.cpp:995-1011);
             takes ~4 minutes, which is why the keep bits are stored instead of recomputed in the GPU test.
  nms2d_2048: S2D-uniform 2048^2 (416 700 -> 25 628), threshold 0.4 (also recomputed live by the GPU test; stored as a second pin).

usage (where /root/reference exists and `make -C oracle ref` has run):  python tests/golden/make_fullsize_golden.py
writes tests/golden/fullsize_keep.npz (packed bits)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref, synth                      # noqa: E402
from stardist_amd.rays3d import Rays_GoldenSpiral  # noqa: E402

out = {}
d, p, s = synth.s2d_uniform(2048, 2048)
t = time.time()
k2 = ref.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(0.4))
print("2D 2048^2: %d -> %d  (%.1f s)" % (len(d), k2.sum(), time.time() - t), flush=True)
out["nms2d_2048_n"] = np.int64(len(d)); out["nms2d_2048_keep"] = np.packbits(k2)

rays = Rays_GoldenSpiral(96)
V, F = rays.vertices, rays.faces.astype(np.int32)
d, p, s, nobj = synth.s3d_nuclei(256, V)
m3 = ref.stardist3d()
ref.set_threads(1)
t = time.time()
k3 = m3.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(0.3))
print("3D 256^3: %d -> %d  (%.1f s, 1 thread)" % (len(d), k3.sum(), time.time() - t), flush=True)
out["nms3d_256_n"] = np.int64(len(d)); out["nms3d_256_keep"] = np.packbits(k3)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fullsize_keep.npz"), **out)
print("written")
