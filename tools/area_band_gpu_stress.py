"""The decision band of the 2D NMS against the Clipper-exact sweep ON THE DEVICE, at scale: random star-polygon pairs of the eleven families of
tests/test_gpu_parity2d.py::test_area_enclosure_contains_clipper_area, fresh seeds, until the time budget is spent.  Per family: pairs, usable
pairs, worst |A_clipper - A| / band.  usage: python tools/area_band_gpu_stress.py [seconds] [first seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_parity2d import _star_polys
from stardist_amd.lib import stardist2d as sd2

FAM = [(32, 10, 0.1, 12, 0.8), (32, 10, 0.03, 6, 1.0), (32, 10, 0.03, 3, 0.97), (32, 20, 0.05, 6, 0.95), (32, 10, 0.3, 25, 0.8), (32, 4, 0.3, 6, 0.8),
       (32, 2.5, 0.3, 4, 1.0), (16, 25, 0.2, 30, 0.8), (32, 40, 0.1, 60, 0.9), (32, 10, 0.9, 12, 0.8), (24, 200, 0.2, 300, 0.8)]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
acc = [[0, 0, 0.0, 0.0] for _ in FAM]                 # pairs, usable, worst ratio, worst |d|
t0 = time.time(); rnd = 0
while time.time() - t0 < budget:
    for fi, (R, radius, noise, spread, scale) in enumerate(FAM):
        rng = np.random.RandomState(seed0 + 1000 * rnd + fi)
        n = 400000
        xa, ya = _star_polys(rng, n, R, radius, noise, spread)
        xb, yb = _star_polys(rng, n, R, radius * scale, noise, spread)
        twice, flags = sd2.clip_pairs(xa, ya, xb, yb)
        ok = (flags & 0xFF) == 0
        area, band, usable, K, T = sd2.area_bounds_pairs(xa, ya, xb, yb)
        us = usable & ok
        d = np.abs(0.5 * twice.astype(np.float64) - area.astype(np.float64))
        a = acc[fi]
        a[0] += n; a[1] += int(us.sum())
        if us.any():
            r = d[us] / band[us]
            a[2] = max(a[2], float(r.max())); a[3] = max(a[3], float(d[us].max()))
        if time.time() - t0 >= budget: break
    rnd += 1
print("area enclosure vs the Clipper-exact sweep on the device: %d rounds of 400 000 pairs per family, seeds from %d, %.0f s" % (rnd, seed0, time.time() - t0))
tot = 0; tu = 0; w = 0.0
for (R, radius, noise, spread, scale), a in zip(FAM, acc):
    print("  R=%d radius=%g noise=%g spread=%g scale=%g: %d pairs, %d usable, worst |A_clipper - A| / band %.4f (max |d| %.2f)" % (R, radius, noise, spread, scale, a[0], a[1], a[2], a[3]))
    tot += a[0]; tu += a[1]; w = max(w, a[2])
print("total %d pairs, %d usable, worst %.4f of the band (a violation would be > 1)" % (tot, tu, w))
