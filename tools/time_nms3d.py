"""Time the 3D NMS alone on the S3D-nuclei generator (tests' full-size workload). Usage: python tools/time_nms3d.py [size] [reps]"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import synth
from stardist_amd.lib import _native, stardist3d as sd3
from stardist_amd.rays3d import Rays_GoldenSpiral

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rays = Rays_GoldenSpiral(96)
V, F = rays.vertices, rays.faces.astype(np.int32)
d, p, s, nobj = synth.s3d_nuclei(size, V)
dev = torch.device('cuda:0')
td, tp, ts, tV, tF = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (d, p, s, V.astype(np.float32), F))
for r in range(reps):
    t = time.time()
    keep = sd3.c_non_max_suppression_inds(td, tp, tV, tF, ts, 1, 1, 0, np.float32(0.3))
    torch.cuda.synchronize()
    dt = time.time() - t
    st = _native.last_stats["nms3d"]
    print(f"rep {r}: N={len(d)} -> {int(keep.sum())}  {dt*1e3:.1f} ms  stage3 {st[8]/1e6:.1f} ms ({st[2]} pairs)  stage4 {st[9]/1e6:.1f} ms ({st[11]} pairs)  "
          f"stage5 {st[10]/1e6:.1f} ms ({st[3]})  rounds {st[4]}  near-threshold exact-volume decisions {st[13]} large-face fallbacks {st[14]}  |  broad phase (precompute + grid + neighbour lists, "
          f"{st[5]} list entries) {st[15]/1e6:.2f} ms = {401.0 * len(d) / max(1, st[15]):.1f} GB/s of the 401 B/candidate the scan must move", flush=True)
