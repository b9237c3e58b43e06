"""How long does ONE Clipper-exact sweep take on the device, and how does a launch scale with the number of pairs?  The pair-level probe
(sd_clip_pairs_device: AddPath of both polygons + the bound-slot sweep per lane, tier 1) on n = 1 ... 2^17 pairs of bench-like star polygons
(32 rays, radius ~ 14, overlap near the NMS threshold).  usage: python tools/time_clip_latency.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stardist_amd.lib import _native as N

rng = np.random.default_rng(0)
R = 32
phi = 2 * np.pi * np.arange(R) / R
dev = torch.device("cuda:0")
NMAX = 1 << 17


def polys(n, shift):
    d = 14.0 * (1 + 0.15 * rng.standard_normal((n, R))).astype(np.float32)
    c = rng.uniform(200, 210, (n, 2)).astype(np.float32) + shift
    x = (c[:, 1:2] + d * np.cos(phi)).astype(np.int32); y = (c[:, 0:1] + d * np.sin(phi)).astype(np.int32)
    return x, y


xa, ya = polys(NMAX, 0.0); xb, yb = polys(NMAX, np.float32([9.0, 6.0]))
T = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (xa, ya, xb, yb)]
out = torch.zeros(NMAX, dtype=torch.int64, device=dev); fl = torch.zeros(NMAX, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for n in (1, 1, 4, 16, 64, 256, 1024, 4096, 16384, 65536, 98304, NMAX):
    best = 1e9
    for rep in range(5):
        e0.record()
        N.check(N.lib().sd_clip_pairs_device(N.tptr(T[0]), N.tptr(T[1]), N.tptr(T[2]), N.tptr(T[3]), n, R, N.tptr(out), N.tptr(fl), N.current_stream()))
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print("n = %6d pairs: %.3f ms  (%.1f ns/pair)" % (n, best, best * 1e6 / n), flush=True)
