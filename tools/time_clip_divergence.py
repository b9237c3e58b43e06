"""How much of the pair kernel's time is lane divergence?  Times the pair-level probe (same scan-beam sweep) on
(a) 2^18 DIFFERENT random star-polygon pairs and (b) 2^18 copies of ONE pair (all 64 lanes of a wave follow the same path).
usage: python tools/time_clip_divergence.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stardist_amd.lib import _native as N

rng = np.random.default_rng(0)
n, R = 1 << 18, 32
phi = 2 * np.pi * np.arange(R) / R


def polys(n):
    d = 10.0 * (1 + 0.1 * rng.standard_normal((n, R))).astype(np.float32)
    c = rng.uniform(100, 110, (n, 2)).astype(np.float32)
    x = (c[:, 1:2] + d * np.cos(phi)).astype(np.int32); y = (c[:, 0:1] + d * np.sin(phi)).astype(np.int32)
    return x, y


xa, ya = polys(n); xb, yb = polys(n)
dev = torch.device("cuda:0")


def run(tag, arrs):
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]
    out = torch.zeros(n, dtype=torch.int64, device=dev); fl = torch.zeros(n, dtype=torch.int32, device=dev)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        N.check(N.lib().sd_clip_pairs_device(N.tptr(t[0]), N.tptr(t[1]), N.tptr(t[2]), N.tptr(t[3]), n, R, N.tptr(out), N.tptr(fl), N.current_stream()))
        torch.cuda.synchronize(); dt = time.time() - t0
    f = fl.cpu().numpy()
    print(f"{tag}: {dt*1e3:.2f} ms for {n} pairs = {dt/n*1e9:.1f} ns/pair, general path {int((f & 256 != 0).sum())}, capacity flags {int(((f & 512 != 0) & (f & (1|2|32|64|128) != 0)).sum())}", flush=True)


run("different pairs per lane", (xa, ya, xb, yb))
one = [np.repeat(a[:1], n, axis=0) for a in (xa, ya, xb, yb)]
run("same pair in every lane ", one)
# same pair per WAVE (64 consecutive identical), different across waves
idx = (np.arange(n) // 64) * 64
run("same pair per wave      ", [a[idx] for a in (xa, ya, xb, yb)])
