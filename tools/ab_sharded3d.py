"""A/B of the sharded 3D leg (one 1024^3 volume) with a probe library: STARDIST_AMD_PROBE_LIB=NAME python tools/ab_sharded3d.py"""
import _probe_lib  # noqa: F401  (first: selects the library)
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import synth
from stardist_amd.models import Config3D, StarDist3D
dev = torch.device("cuda:0")
S = 256
vol = torch.from_numpy(synth.s3d_nuclei_image(S, seed=0)).to(dev)
m3 = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
m3.thresholds = dict(prob=0.5, nms=0.3)
bench.calibrate_heads(m3, vol, frac=0.009, radius=8.5, noise=0.03)
base3 = synth.s3d_nuclei_image(S, seed=0)
bigv = bench.sharded_input(m3, base3, 4, "ZYX", 560, 32, 32, 0, 1, dev)
for rep in range(2):
    r = bench.run_sharded_leg(m3, bigv, "ZYX", 560, 32, 32, 1, 1, None, 0)
    print("RESULT", os.environ.get("STARDIST_AMD_PROBE_LIB", "product"), json.dumps({k: r[k] for k in r if k in ("value", "s_per_pass", "t_predict", "t_local_nms", "t_final", "instances")}), flush=True)
