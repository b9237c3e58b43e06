"""Time the 2D NMS alone on S2D-uniform. Usage: python tools/time_nms2d.py [size] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import synth
from stardist_amd.lib import _native, stardist2d as sd2

size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d, p, s = synth.s2d_uniform(size, size)
dev = torch.device("cuda:0")
td, tp = torch.from_numpy(d).to(dev), torch.from_numpy(p).to(dev)
for r in range(reps):
    torch.cuda.synchronize(); t = time.time()
    keep = sd2.c_non_max_suppression_inds(td, tp, 1, 1, 0, np.float32(0.4))
    torch.cuda.synchronize(); dt = time.time() - t
    st = _native.last_stats["nms2d"]
    print(f"rep {r}: N={len(d)} -> {int(keep.sum())}  {dt*1e3:.1f} ms  pair kernel {st[4]/1e6:.2f} ms ({st[0]} pairs, {st[5]} launches)  join {st[6]/1e6:.2f} ms ({st[1]})  "
          f"build {st[7]/1e6:.2f} ms  rounds {st[2]}  nbr {st[3]}", flush=True)
